"""CPU oracle for the AVLMaps hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (avlmaps_amd) must never import it.
"""

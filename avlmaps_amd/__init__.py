"""avlmaps_amd -- MI355X (gfx950) implementation of the AVLMaps map-creation / landmark-indexing hot path.

Compute lives in the hand-written HIP library libavlmaps_hip.so (C ABI: include/avlmaps_hip.h); this
package is the thin ctypes layer plus a host-side mirror of the reference's builder / indexer interface
(avlmaps.map.VLMap, VLMapBuilder, avlmaps.utils.clip_utils.get_lseg_score, ...).  There is no CPU
fallback: every compute entry point raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"

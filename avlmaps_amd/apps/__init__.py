"""Command-line counterparts of the reference's application/create_map.py and application/index_map.py (object branch)."""

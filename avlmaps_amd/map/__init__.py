"""Host-side mirror of the reference's avlmaps.map builder / indexer interface (VLMap, VLMapBuilder, Map, AVLMap)."""
from .map import Map  # noqa: F401
from .vlmap import VLMap  # noqa: F401
from .vlmap_builder import VLMapBuilder  # noqa: F401
from .avlmap import AVLMap  # noqa: F401
from .vlmap_builder_multi_floor import VLMapBuilderMultiFloor  # noqa: F401
from .vlmap_multi_floor import VLMapMultiFloor  # noqa: F401

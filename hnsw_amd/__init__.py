"""hnsw_amd: MI355X-native HNSW query engine.  `HNSW` mirrors the reference class."""
from .flatgraph import FlatGraph  # noqa: F401
from .index import HNSW, Neighbors  # noqa: F401

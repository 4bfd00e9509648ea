"""Category matching helper with the reference's name (avlmaps/utils/index_utils.py:8-32).

Upstream asks an OpenAI model to pick the closest category; that network call is outside the hot path, so
this mirror resolves exact / case-insensitive / substring matches locally and otherwise raises."""
from __future__ import annotations

from typing import List


def find_similar_category_id(class_name: str, classes_list: List[str]) -> int:
    if class_name in classes_list:
        return classes_list.index(class_name)
    low = [c.lower() for c in classes_list]
    name = class_name.lower().strip()
    if name in low:
        return low.index(name)
    hits = [i for i, c in enumerate(low) if name in c or c in name]
    if len(hits) == 1:
        return hits[0]
    raise KeyError(f"{class_name!r} does not match one of the initialised categories {classes_list}; "
                   "upstream delegates this to an LLM (index_utils.py:8-32), which is out of scope here")

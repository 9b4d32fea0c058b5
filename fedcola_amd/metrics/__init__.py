"""Mirror of the reference's src/metrics package for the retrieval-evaluation row."""
from .eval_coco import COCOEvaluator, recall_at_k  # noqa: F401

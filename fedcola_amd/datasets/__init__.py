"""On-disk dataset formats of the retrieval clients."""
from .coco import CocoCaptionsCap, fetch_coco, public_set  # noqa: F401
from .flickr30k import Flickr30kCap, fetch_flickr30k  # noqa: F401

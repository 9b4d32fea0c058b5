"""Input side of the client step: client splits (bit-exact with the reference's numpy RNG stream) and the
device prefetcher that hands batches to the fused step."""
from .split import simulate_split  # noqa: F401
from .batch import PinnedBatchLoader  # noqa: F401
from .prefetch import DevicePrefetcher  # noqa: F401
from .cache import DecodedCache  # noqa: F401
from .tokenizer import BertVocabTokenizer, build_tokenizer  # noqa: F401

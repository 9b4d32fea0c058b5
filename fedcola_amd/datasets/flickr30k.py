"""``Flickr30kCap`` (/root/reference/src/datasets/flickr30k.py:9-45): the '|'-delimited ``{split}.csv`` annotation table
(columns ``image_name| comment_number| comment``, five consecutive rows per image) over ``<root>/flickr30k_images/``.

Sample tuple = (image, token ids, image id = index // 5, annotation id = index, index) -- what the retrieval client and
``COCOEvaluator.extract_features`` unpack (fedavgclient.py:91, eval_coco.py:176)."""
from __future__ import annotations

import logging
import os

import pandas as pd
from torch.utils.data import Dataset

logger = logging.getLogger(__name__)


class Flickr30kCap(Dataset):
    def __init__(self, root, split="train", transform=None, tokenizer=None, max_length=40, train_all=False):
        self.root, self.split, self.transform = root, split, transform
        anno = pd.read_csv(os.path.join(root, f"{'train_all' if train_all else split}.csv"), delimiter="|")
        self.images = anno["image_name"]
        self.captions = [str(c) for c in anno[" comment"].values]     # the header keeps its leading blank
        self.tokenizer, self.max_length = tokenizer, max_length
        self.n_images = len(set(self.images))
        self.iid_to_cls = {}

    def _caption(self, index):
        caption = self.captions[index]
        if self.tokenizer is not None:
            caption = self.tokenizer(caption, padding="max_length", truncation=True, max_length=self.max_length,
                                     return_tensors="pt")["input_ids"][0]
        return caption

    def __getitem__(self, index):
        from PIL import Image
        image = Image.open(os.path.join(self.root, "flickr30k_images", self.images[index])).convert("RGB")
        if self.transform is not None:
            image = self.transform(image)
        return image, self._caption(index), index // 5, index, index

    # ---- what loaders.cache.DecodedCache asks of a caption dataset: which image a sample shows (five captions share one), and the
    # sample's other fields without decoding the image
    def image_key(self, index):
        return self.images[index]

    def sample_without_image(self, index):
        return None, self._caption(index), index // 5, index, index

    def __len__(self):
        return len(self.images)


def fetch_flickr30k(args, root, transforms, tokenizer, modality="img+txt"):
    """flickr30k.py:47-80: train / test instances tagged with task / modality / name; sets args.in_channels, args.num_classes."""
    kw = dict(root=root, transform=transforms[0], split="train", tokenizer=tokenizer, max_length=args.seq_len,
              train_all=args.flickr_train_all)
    raw_train = Flickr30kCap(**kw)
    kw.update(transform=transforms[1], split="test", train_all=False)
    raw_test = Flickr30kCap(**kw)
    for d in (raw_train, raw_test):
        d.task, d.modality, d.name = "img+txt", modality, "Flickr30k"
    args.in_channels = 3
    args.num_classes = None
    return raw_train, raw_test, args

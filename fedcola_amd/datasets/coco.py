"""``CocoCaptionsCap`` (/root/reference/src/datasets/coco.py:25-159) and ``fetch_coco`` (:192-225): COCO caption annotations
(``captions_{train,val}2014.json``), caption-id lists (``coco_{train,test}_ids.npy``, int64 annotation ids) and the optional
instance-annotation class map (``iid_to_cls``) over ``<root>/all_images``.

The reference indexes the annotation file with pycocotools' ``COCO`` (not installed here: **parity unpinned** at that boundary).
Only its index is needed -- ``anns[id]`` and ``imgs[id]`` of the published annotation format -- and is rebuilt from the JSON
directly (``_CocoIndex``); sample tuple = (image, token ids, image_id, annotation_id, index), as the retrieval clients, the
evaluator and CreamFL's public loader unpack it."""
from __future__ import annotations

import json
import logging
import operator
import os
from glob import glob

import numpy as np
from torch.utils.data import Dataset

logger = logging.getLogger(__name__)


class _CocoIndex:
    """anns / imgs dictionaries of a COCO annotation dict (what pycocotools.coco.COCO.createIndex builds)."""

    def __init__(self, dataset: dict):
        self.dataset = dataset
        self.anns = {a["id"]: a for a in dataset.get("annotations", [])}
        self.imgs = {i["id"]: i for i in dataset.get("images", [])}

    def loadAnns(self, ids):
        return [self.anns[i] for i in (ids if isinstance(ids, (list, tuple)) else [ids])]

    def loadImgs(self, ids):
        return [self.imgs[i] for i in (ids if isinstance(ids, (list, tuple)) else [ids])]


def _load(path):
    with open(path, "r") as f:
        d = json.load(f)
    if not isinstance(d, dict):
        raise TypeError("invalid type {}".format(type(d)))
    return d


class CocoCaptionsCap(Dataset):
    def __init__(self, root, annFile, ids=None, extra_annFile=None, extra_ids=None, transform=None, target_transform=None, tokenizer=None,
                 max_length=40, instance_annFile=None, client=-1):
        self.root = os.path.expanduser(root)
        dataset = _load(annFile)
        if extra_annFile:                                              # coco.py:65-79: merged train + extra annotations
            extra = _load(extra_annFile)
            if set(dataset.keys()) != set(extra.keys()):
                raise KeyError("key mismatch {} != {}".format(list(dataset.keys()), list(extra.keys())))
            for key in ("images", "annotations"):
                dataset[key].extend(extra[key])
        self.coco = _CocoIndex(dataset)
        self.ids = list(self.coco.anns.keys()) if ids is None else list(ids)
        if extra_ids is not None:
            self.ids += list(extra_ids)
        self.ids = [int(i) for i in self.ids]
        self.transform, self.target_transform, self.tokenizer, self.max_length = transform, target_transform, tokenizer, max_length
        self.all_image_ids = set(self.coco.anns[a]["image_id"] for a in self.ids)
        iid_to_cls = {}
        if instance_annFile:                                           # coco.py:93-121: 90-way category code -> dense class id
            for ins_file in glob(instance_annFile + "/instances_*"):
                for ann in _load(ins_file)["annotations"]:
                    image_id = int(ann["image_id"])
                    code = iid_to_cls.get(image_id, [0] * 90)
                    code[int(ann["category_id"]) - 1] = 1
                    iid_to_cls[image_id] = code
                seen, dense = {}, {}
                for k, v in iid_to_cls.items():                        # (the reference re-maps after every file, on the mapped values)
                    key = "".join(str(s) for s in v) if isinstance(v, list) else str(v)
                    if key not in seen:
                        seen[key] = len(seen)
                    dense[k] = seen[key]
                iid_to_cls = dense
                missing = self.all_image_ids - set(iid_to_cls.keys())
                if missing:
                    print(f"Found mismatched! {len(missing)}")
        self.iid_to_cls = iid_to_cls
        self.n_images = len(self.all_image_ids)

    def reduce_samples(self, num_samples=1000):
        """coco.py:123-128: keep the LAST num_samples caption ids."""
        self.ids = list(operator.itemgetter(*np.arange(-num_samples, 0))(self.ids))

    def _target(self, caption):
        if self.tokenizer is not None:
            return self.tokenizer(caption, padding="max_length", truncation=True, max_length=self.max_length, return_tensors="pt")["input_ids"][0]
        return caption

    def __getitem__(self, index):
        from PIL import Image
        annotation_id = self.ids[index]
        annotation = self.coco.anns[annotation_id]
        image_id, caption = annotation["image_id"], annotation["caption"]
        img = Image.open(os.path.join(self.root, self.coco.imgs[image_id]["file_name"])).convert("RGB")
        if self.transform is not None:
            img = self.transform(img)
        return img, self._target(caption), image_id, annotation_id, index

    # ---- what loaders.cache.DecodedCache asks of a caption dataset (see datasets/flickr30k.py)
    def image_key(self, index):
        return self.coco.anns[self.ids[index]]["image_id"]

    def sample_without_image(self, index):
        annotation_id = self.ids[index]
        annotation = self.coco.anns[annotation_id]
        return None, self._target(annotation["caption"]), annotation["image_id"], annotation_id, index

    def __len__(self):
        return len(self.ids)


def fetch_coco(args, root, transforms, tokenizer, modality="img+txt"):
    """coco.py:192-225: train = captions_train2014 restricted to coco_train_ids.npy[:reduce_samples], test = captions_val2014 with
    coco_test_ids.npy."""
    img_path = os.path.join(root, "all_images")
    kw = dict(root=img_path, annFile=os.path.join(root, "annotations", "captions_train2014.json"), transform=transforms[0], tokenizer=tokenizer,
              max_length=args.seq_len, ids=np.load(os.path.join(root, "coco_train_ids.npy"))[:args.reduce_samples])
    raw_train = CocoCaptionsCap(**kw)
    kw.update(transform=transforms[1], annFile=os.path.join(root, "annotations", "captions_val2014.json"),
              ids=np.load(os.path.join(root, "coco_test_ids.npy")))
    raw_test = CocoCaptionsCap(**kw)
    for d in (raw_train, raw_test):
        d.task, d.modality, d.name = "img+txt", modality, "Coco"
    args.in_channels = 3
    args.num_classes = None
    return raw_train, raw_test, args


def public_set(root, anno_path, num_pub_samples, transform=None, tokenizer=None, max_length=40):
    """CreamflServer.get_pub_loader's dataset (creamflserver.py:100-113): the LAST num_pub_samples ids of coco_train_ids.npy, which sits
    two directories above the annotation file."""
    parent = os.sep.join(anno_path.split("/")[:-2])
    ids = np.load(os.path.join(parent, "coco_train_ids.npy"))[-num_pub_samples:]
    return CocoCaptionsCap(root, anno_path, transform=transform, tokenizer=tokenizer, max_length=max_length, ids=ids)

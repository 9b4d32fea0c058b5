"""BERT word-piece tokenisation over a plain ``vocab.txt`` -- what the reference builds with
``BertTokenizer(os.path.join(args.data_path, 'vocab.txt'))`` (/root/reference/src/loaders/data.py:182-190, VOCABS :28-31) and
calls as ``tokenizer(caption, padding='max_length', truncation=True, max_length=L, return_tensors='pt')['input_ids'][0]``
(src/datasets/flickr30k.py:39-40, src/datasets/coco.py:151-152) or through ``partial(tokenizer, padding='max_length',
max_length=seq_len, truncation=True)`` (data.py:299-303).

The algorithm is the published BERT one (basic tokenisation: clean, whitespace split, lower-case + accent stripping, punctuation
split, CJK isolation; then greedy longest-match word pieces with the ``##`` continuation prefix; ``[CLS] ... [SEP]``, truncation to
max_length, ``[PAD]`` to max_length).  Host-side input plumbing of row N4: ids are bit-exact against HF's BertTokenizer on the
reference's Flickr30k vocabulary (tests/test_data_golden.py, golden ids generated with transformers in the build container)."""
from __future__ import annotations

import os
import unicodedata
from typing import Dict, List, Optional, Sequence, Union


def load_vocab(path: str) -> Dict[str, int]:
    vocab: Dict[str, int] = {}
    with open(path, "r", encoding="utf-8") as f:
        for i, line in enumerate(f):
            vocab[line.rstrip("\n")] = i
    return vocab


def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F
            or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class BertVocabTokenizer:
    def __init__(self, vocab_file: str, do_lower_case: bool = True, unk_token="[UNK]", sep_token="[SEP]", pad_token="[PAD]", cls_token="[CLS]",
                 mask_token="[MASK]", max_input_chars_per_word: int = 100):
        if not os.path.isfile(vocab_file):
            raise ValueError(f"Can't find a vocabulary file at path '{vocab_file}'.")
        self.vocab = load_vocab(vocab_file)
        self.ids_to_tokens = {i: t for t, i in self.vocab.items()}
        self.do_lower_case = do_lower_case
        self.unk_token, self.sep_token, self.pad_token, self.cls_token, self.mask_token = unk_token, sep_token, pad_token, cls_token, mask_token
        self.never_split = {unk_token, sep_token, pad_token, cls_token, mask_token}
        self.max_chars = max_input_chars_per_word
        self.unk_token_id = self.vocab.get(unk_token)
        self.pad_token_id = self.vocab.get(pad_token, 0)
        self.cls_token_id = self.vocab.get(cls_token, self.unk_token_id)
        self.sep_token_id = self.vocab.get(sep_token, self.unk_token_id)

    @property
    def vocab_size(self) -> int:
        return len(self.vocab)

    def __len__(self):
        return len(self.vocab)

    # ---- basic tokenisation
    def _basic(self, text: str) -> List[str]:
        out = []
        for ch in text:                                   # clean: drop NUL / replacement / control characters, whitespace -> blank
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_cjk(cp):
                out.append(f" {ch} ")
            else:
                out.append(" " if _is_whitespace(ch) else ch)
        text = unicodedata.normalize("NFC", "".join(out))
        tokens: List[str] = []
        for tok in text.strip().split():
            if tok not in self.never_split:
                if self.do_lower_case:
                    tok = tok.lower()
                    tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
                tokens.extend(self._split_punct(tok))
            else:
                tokens.append(tok)
        return " ".join(tokens).split()

    @staticmethod
    def _split_punct(tok: str) -> List[str]:
        pieces: List[List[str]] = []
        new_word = True
        for ch in tok:
            if _is_punctuation(ch):
                pieces.append([ch])
                new_word = True
            else:
                if new_word:
                    pieces.append([])
                new_word = False
                pieces[-1].append(ch)
        return ["".join(p) for p in pieces]

    # ---- word pieces: greedy longest match first
    def _wordpiece(self, token: str) -> List[str]:
        if len(token) > self.max_chars:
            return [self.unk_token]
        pieces, start = [], 0
        while start < len(token):
            end, cur = len(token), None
            while start < end:
                sub = token[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    cur = sub
                    break
                end -= 1
            if cur is None:
                return [self.unk_token]
            pieces.append(cur)
            start = end
        return pieces

    def tokenize(self, text: str) -> List[str]:
        out: List[str] = []
        for tok in self._basic(text):
            if tok in self.never_split:
                out.append(tok)
            else:
                out.extend(self._wordpiece(tok))
        return out

    def convert_tokens_to_ids(self, tokens: Sequence[str]) -> List[int]:
        return [self.vocab.get(t, self.unk_token_id) for t in tokens]

    def convert_ids_to_tokens(self, ids: Sequence[int]) -> List[str]:
        return [self.ids_to_tokens.get(int(i), self.unk_token) for i in ids]

    def encode(self, text: str, max_length: Optional[int] = None, truncation: bool = False, padding: Union[bool, str, None] = None) -> List[int]:
        ids = self.convert_tokens_to_ids(self.tokenize(text))
        if truncation and max_length is not None and len(ids) > max_length - 2:
            ids = ids[: max(max_length - 2, 0)]
        ids = [self.cls_token_id] + ids + [self.sep_token_id]
        if padding == "max_length" and max_length is not None and len(ids) < max_length:
            ids = ids + [self.pad_token_id] * (max_length - len(ids))
        return ids

    def __call__(self, text: Union[str, Sequence[str]], padding=None, truncation=False, max_length=None, return_tensors=None, **unused):
        """The subset of the HF call the reference uses.  One string -> dict of lists (or [1, L] tensors with return_tensors='pt');
        a list of strings -> lists of lists ([n, L] tensors; padding=True / 'longest' pads to the longest)."""
        single = isinstance(text, str)
        rows = [self.encode(t, max_length, truncation, padding if padding == "max_length" else None) for t in ([text] if single else text)]
        if padding in (True, "longest"):
            L = max(len(r) for r in rows)
            rows = [r + [self.pad_token_id] * (L - len(r)) for r in rows]
        mask = [[int(i != self.pad_token_id or j < 1) for j, i in enumerate(r)] for r in rows]
        for r, m in zip(rows, mask):          # attention mask: 1 up to and including [SEP]
            n = len(r)
            while n > 0 and r[n - 1] == self.pad_token_id:
                n -= 1
            m[:] = [1] * n + [0] * (len(r) - n)
        types = [[0] * len(r) for r in rows]
        if return_tensors == "pt":
            import torch
            return {"input_ids": torch.tensor(rows, dtype=torch.int64), "token_type_ids": torch.tensor(types, dtype=torch.int64),
                    "attention_mask": torch.tensor(mask, dtype=torch.int64)}
        if single:
            return {"input_ids": rows[0], "token_type_ids": types[0], "attention_mask": mask[0]}
        return {"input_ids": rows, "token_type_ids": types, "attention_mask": mask}


VOCABS = {"Flickr30k": "vocab.txt", "MedicalAbstracts": "vocab.txt"}      # data.py:28-31


def build_tokenizer(args):
    """data.py:173-190 for ``--use_bert_tokenizer``: datasets with their own vocabulary file get a tokenizer over it (the model's
    ``vocab_size`` then equals its length: 7 732 for Flickr30k, fedavgserver.py:89-92); others need HF's pretrained 'bert-base-uncased'
    files, which this build cannot fetch."""
    if not getattr(args, "use_bert_tokenizer", False):
        return None
    if args.dataset in VOCABS:
        return BertVocabTokenizer(os.path.join(args.data_path, VOCABS[args.dataset]))
    raise NotImplementedError("BertTokenizer.from_pretrained('bert-base-uncased') needs files that are not available offline; "
                              "point args.data_path at a directory with vocab.txt and add the dataset to VOCABS")

"""Generates tests/golden/tokenizer.json in the build container: token ids of HF's BertTokenizer (the class the reference builds at
/root/reference/src/loaders/data.py:182-190) over the reference's Flickr30k vocabulary (data/flickr30k/vocab.txt, copied as the data
fixture tests/golden/flickr30k_vocab.txt) for hand-picked and seeded random sentences at max_length 32 / 40 / 8.
    python tests/golden/make_tokenizer_golden.py"""
import json
import os
import random
import shutil

from transformers import BertTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))
VOC = os.path.join(HERE, "flickr30k_vocab.txt")
if os.path.exists("/root/reference/data/flickr30k/vocab.txt"):
    shutil.copyfile("/root/reference/data/flickr30k/vocab.txt", VOC)
hf = BertTokenizer(VOC)
words = [l.rstrip("\n") for l in open(VOC)]
rng = random.Random(0)
sents = ["Two young guys with shaggy hair look at their hands while hanging out in the yard.",
         "A man in a blue shirt, standing on a ladder -- cleaning a window!",
         "Café naïve résumé ÀÉÎ zzzqqq unbelievablewordthatisnotinvocab",
         "several men in hard hats are operating a giant pulley system .",
         "x" * 120 + " dog", "", "   ", "dog's ball; (red) [blue] {green} #tag @user 50% a+b=c", "日本語 の text 中文字",
         "[CLS] hello [SEP] [MASK] [PAD] [UNK]", "A\tman\nrides\ra  bike fast​.", "snowboarding skateboarder's playgrounds"]
for _ in range(60):
    toks = []
    for _ in range(rng.randint(1, 45)):
        r = rng.random()
        w = rng.choice(words)
        if r < 0.1:
            w = w.upper()
        elif r < 0.15:
            w = w + rng.choice(["ing", "s", "ed", "xyz"])
        elif r < 0.2:
            w = rng.choice([",", ".", "!", "?", "-", "'"])
        toks.append(w)
    sents.append(" ".join(toks))
recs = [dict(text=s, max_length=L, ids=hf(s, padding="max_length", truncation=True, max_length=L, return_tensors="pt")["input_ids"][0].tolist())
        for s in sents for L in (32, 40, 8)]
json.dump(recs, open(os.path.join(HERE, "tokenizer.json"), "w"))
print(len(recs), "cases")

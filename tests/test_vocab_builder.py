"""Vocabulary: the interface contract of vocab_builder.py:11-102.  The nltk tokeniser has no oracle
offline (parity unpinned); what is checked is the id assignment, thresholding, the <unk> fallback and the file round trip."""
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from showtell_amd.vocab_builder import DatasetVocabulary, encode_caption, get_vocabulary, tokenize  # noqa: E402


def test_tokenize_caption_rules():
    assert tokenize("a man's dog isn't on the beach.") == ["a", "man", "'s", "dog", "is", "n't", "on", "the", "beach", "."]
    assert tokenize("two dogs, one red ball") == ["two", "dogs", ",", "one", "red", "ball"]
    assert tokenize("a black-and-white cat") == ["a", "black-and-white", "cat"]
    assert tokenize("") == []


def test_get_vocabulary_mscoco_and_flickr(tmp_path):
    caps = ["A dog runs.", "a dog sits", "A cat sits on a mat", "the dog"]
    ann = {"annotations": [{"id": i, "image_id": i, "caption": c} for i, c in enumerate(caps)]}
    (tmp_path / "ann.json").write_text(json.dumps(ann))
    params = {"vocab_path": str(tmp_path / "vocab.json"), "data_dir": str(tmp_path), "train_ann_path": "ann.json", "vocab_threshold": 2}
    v = get_vocabulary("MSCOCO", params)
    # ids 0..3 are the keywords (vocab_builder.py:68-69); then words with count >= 2 in order of first appearance
    assert [v.index_to_word[i] for i in range(len(v))] == ["<pad>", "<start>", "<end>", "<unk>", "a", "dog", "sits"]
    assert v("dog") == 5 and v("zebra") == v("<unk>") == 3 and len(v) == 7
    assert v.start_token() == "<start>" and v.end_token() == "<end>"
    assert encode_caption(v, "A dog flies") == [1, 4, 5, 3, 2]
    again = get_vocabulary("MSCOCO", params)                 # second call loads the saved file
    assert again.word_to_index == v.word_to_index and again.index == v.index
    (tmp_path / "fl.tsv").write_text("1.jpg#0\tA dog runs .\n1.jpg#1\ta dog\n")
    p2 = dict(params, vocab_path=str(tmp_path / "v2.json"), train_ann_path="fl.tsv", vocab_threshold=1)
    assert [w for w in get_vocabulary("Flickr", p2).word_to_index][4:] == ["a", "dog", "runs", "."]
    with pytest.raises(ValueError):
        get_vocabulary("ImageNet", dict(params, vocab_path=str(tmp_path / "none.json")))


def test_vocabulary_feeds_caption_word_format():
    from showtell_amd.utils import create_caption_word_format
    v = DatasetVocabulary()
    for w in ["<pad>", "<start>", "<end>", "<unk>", "a", "dog"]:
        v.add_new_word(w)
    v.add_new_word("dog")
    assert len(v) == 6
    assert create_caption_word_format([[1, 4, 5, 2, 0, 0]], v, False) == [["a", "dog"]]
    assert create_caption_word_format([[1, 4, 5, 2, 0, 0]], v, True) == [[["a", "dog"]]]

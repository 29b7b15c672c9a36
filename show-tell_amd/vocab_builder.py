"""Vocabulary of the reference's ``vocab_builder.py``.  Host-side Python, no kernels.

``DatasetVocabulary`` keeps the reference's interface (vocab_builder.py:11-44: ``word_to_index``, ``index_to_word``,
``index``, ``add_new_word``, ``__call__`` with the ``<unk>`` fallback, ``__len__``, ``start_token``, ``end_token``) --
it is what ``utils.create_caption_word_format`` and ``RNN_Attn.sentence_index(cnn_feature, vocab)`` read.
``get_vocabulary`` follows vocab_builder.py:46-102: ids 0..3 = ``<pad> <start> <end> <unk>``, then every token whose
count reaches ``params['vocab_threshold']``, in order of first appearance.

Differences, on purpose:
  * the reference tokenises with ``nltk.tokenize.word_tokenize`` and reads COCO through ``pycocotools``; neither is
    installed here or shipped in the reference tree.  COCO's annotation file is plain JSON and is read as such;
    ``tokenize`` below restates the Treebank rules that matter for lower-cased captions (punctuation split off,
    ``n't`` / ``'s`` / ``'re`` ... clitics split).  There is no nltk output to pin it to: **parity unpinned** for the
    tokeniser; pass ``tokenizer=nltk.tokenize.word_tokenize`` where nltk exists.
  * the vocabulary file is JSON (the word list by id), not a pickle of this class.
"""
import json
import os
import re
from collections import Counter

_CLITICS = re.compile(r"(?i)([a-z])(n't|'s|'m|'d|'ll|'re|'ve)\b")
_TOKEN = re.compile(r"n't|'(?:s|m|d|ll|re|ve)\b|\.\.\.|--|[A-Za-z0-9_]+(?:[-.,'][A-Za-z0-9_]+)*|\S")


def tokenize(text):
    """Treebank-style word tokens of one caption (see the module docstring: unpinned)."""
    text = _CLITICS.sub(r"\1 \2", text)
    return _TOKEN.findall(text)


class DatasetVocabulary(object):

    def __init__(self):
        self.word_to_index = {}
        self.index_to_word = {}
        self.index = 0

    def add_new_word(self, word):
        """Adds a new word to the vocabulary (if it doesn't already exist)."""
        if word in self.word_to_index:
            return
        self.word_to_index[word] = self.index
        self.index_to_word[self.index] = word
        self.index += 1

    def __call__(self, word):
        return self.word_to_index.get(word, self.word_to_index['<unk>'])

    def __len__(self):
        return len(self.word_to_index)

    def start_token(self):
        return '<start>'

    def end_token(self):
        return '<end>'

    # -- file format (ours): {"words": [word of id 0, word of id 1, ...]}
    def save(self, path):
        with open(path, 'w') as f:
            json.dump({"words": [self.index_to_word[i] for i in range(self.index)]}, f)

    @classmethod
    def load(cls, path):
        with open(path) as f:
            words = json.load(f)["words"]
        v = cls()
        for w in words:
            v.add_new_word(w)
        return v


def _captions(dataset, annotation_path):
    if dataset == 'MSCOCO':
        with open(annotation_path) as f:
            for ann in json.load(f)['annotations']:          # pycocotools' ``anns`` keeps this order
                yield str(ann['caption'])
    elif dataset == 'Flickr':
        with open(annotation_path) as f:                     # image<TAB>caption per line
            for line in f:
                parts = line.rstrip('\n').split('\t')
                if len(parts) >= 2:
                    yield str(parts[1])
    else:
        raise ValueError("Please specify a valid dataset. %s is invalid." % (dataset))


def get_vocabulary(dataset, params, tokenizer=None):
    """Retrieves the vocabulary for the specified dataset ('MSCOCO' or 'Flickr'); builds and saves it if absent
    (vocab_builder.py:46-102).  ``params``: 'vocab_path', 'data_dir', 'train_ann_path', 'vocab_threshold'."""
    if os.path.isfile(params['vocab_path']):
        print('Loading vocabulary from the existing file.')
        return DatasetVocabulary.load(params['vocab_path'])
    print('Vocabulary does not exist. Creating vocab...')
    if dataset not in ('MSCOCO', 'Flickr'):
        raise ValueError("Please specify a valid dataset. %s is invalid." % (dataset))
    tokenizer = tokenizer or tokenize
    vocab_dataset = DatasetVocabulary()
    for word in ['pad', 'start', 'end', 'unk']:
        vocab_dataset.add_new_word('<' + word + '>')
    caption_tokens = Counter()
    for caption in _captions(dataset, os.path.join(params['data_dir'], params['train_ann_path'])):
        caption_tokens.update(tokenizer(caption.lower()))
    for word, count in caption_tokens.items():
        if count >= params['vocab_threshold']:
            vocab_dataset.add_new_word(word)
    vocab_dataset.save(params['vocab_path'])
    return vocab_dataset


def encode_caption(vocab, caption, tokenizer=None):
    """Token ids of one caption as the reference's dataset builds them (utils.py:49-51):
    ``[<start>] + words + [<end>]``, unknown words -> ``<unk>``."""
    tokens = (tokenizer or tokenize)(str(caption).lower())
    return [vocab(vocab.start_token())] + [vocab(t) for t in tokens] + [vocab(vocab.end_token())]

"""WordPiece tokenisation with the reference's class names (src/lxrt/tokenization.py:72-348).

The reference resolves ``bert-base-uncased`` to a vocabulary URL; offline the vocabulary must
be a local ``vocab.txt`` (one token per line, line number = id).  The algorithm is BERT's
published one: clean + lower-case + strip accents + split punctuation/CJK, then greedy
longest-match-first WordPiece with the ``##`` continuation prefix.
"""
import collections
import os
import unicodedata


def load_vocab(vocab_file):
    vocab = collections.OrderedDict()
    with open(vocab_file, "r", encoding="utf-8") as f:
        for idx, line in enumerate(f):
            tok = line.rstrip("\n")
            if tok:
                vocab[tok.strip()] = idx
    return vocab


def whitespace_tokenize(text):
    text = text.strip()
    return text.split() if text else []


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF
            or 0x2F800 <= cp <= 0x2FA1F)


class BasicTokenizer(object):
    """ref: src/lxrt/tokenization.py:174-288"""

    def __init__(self, do_lower_case=True, never_split=("[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]")):
        self.do_lower_case = do_lower_case
        self.never_split = never_split

    def tokenize(self, text):
        cleaned = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or (unicodedata.category(ch) in ("Cc", "Cf") and ch not in "\t\n\r"):
                continue
            if ch in " \t\n\r" or unicodedata.category(ch) == "Zs":
                cleaned.append(" ")
            elif _is_cjk(cp):
                cleaned.extend((" ", ch, " "))
            else:
                cleaned.append(ch)
        out = []
        for tok in whitespace_tokenize("".join(cleaned)):
            if tok in self.never_split:
                out.append(tok)
                continue
            if self.do_lower_case:
                tok = "".join(c for c in unicodedata.normalize("NFD", tok.lower()) if unicodedata.category(c) != "Mn")
            word = []
            for ch in tok:
                if _is_punct(ch):
                    if word:
                        out.append("".join(word))
                        word = []
                    out.append(ch)
                else:
                    word.append(ch)
            if word:
                out.append("".join(word))
        return out


class WordpieceTokenizer(object):
    """greedy longest-match-first; ref: src/lxrt/tokenization.py:291-348"""

    def __init__(self, vocab, unk_token="[UNK]", max_input_chars_per_word=100):
        self.vocab = vocab
        self.unk_token = unk_token
        self.max_input_chars_per_word = max_input_chars_per_word

    def tokenize(self, text):
        out = []
        for token in whitespace_tokenize(text):
            if len(token) > self.max_input_chars_per_word:
                out.append(self.unk_token)
                continue
            pieces, start, bad = [], 0, False
            while start < len(token):
                end = len(token)
                cur = None
                while start < end:
                    sub = token[start:end]
                    if start > 0:
                        sub = "##" + sub
                    if sub in self.vocab:
                        cur = sub
                        break
                    end -= 1
                if cur is None:
                    bad = True
                    break
                pieces.append(cur)
                start = end
            out.extend([self.unk_token] if bad else pieces)
        return out


class BertTokenizer(object):
    """ref: src/lxrt/tokenization.py:72-171"""

    def __init__(self, vocab_file, do_lower_case=True, max_len=None,
                 never_split=("[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]")):
        if not os.path.isfile(vocab_file):
            raise ValueError("Can't find a vocabulary file at path '{}' (no download offline)".format(vocab_file))
        self.vocab = load_vocab(vocab_file)
        self.ids_to_tokens = collections.OrderedDict([(ids, tok) for tok, ids in self.vocab.items()])
        self.basic_tokenizer = BasicTokenizer(do_lower_case=do_lower_case, never_split=never_split)
        self.wordpiece_tokenizer = WordpieceTokenizer(vocab=self.vocab)
        self.max_len = max_len if max_len is not None else int(1e12)

    def tokenize(self, text):
        return [sub for tok in self.basic_tokenizer.tokenize(text) for sub in self.wordpiece_tokenizer.tokenize(tok)]

    def convert_tokens_to_ids(self, tokens):
        ids = [self.vocab[t] for t in tokens]
        if len(ids) > self.max_len:
            raise ValueError("Token indices sequence length is longer than the specified maximum sequence length "
                             "for this BERT model ({} > {}).".format(len(ids), self.max_len))
        return ids

    def convert_ids_to_tokens(self, ids):
        return [self.ids_to_tokens[i] for i in ids]

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, cache_dir=None, *inputs, **kwargs):
        path = pretrained_model_name_or_path
        if os.path.isdir(path):
            path = os.path.join(path, "vocab.txt")
        return cls(path, *inputs, **kwargs)

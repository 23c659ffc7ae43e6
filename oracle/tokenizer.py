"""Prompt tokenizer restated in plain Python (oracle side).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows helpers/utils.mojo:229-327 (`Tokenizer.__init__`, `find`,
`wrap`, `bpe_encode`) and the file format of tokenizer_creation.py:43-48, with the intended semantics where the literal
code is broken: `str_concat` (:221-231) means concatenation, the id == -1 test precedes the
score load (:303-305), `wrap` maps newline / tab / quotes to their <0xNN> names (:200-209)."""
import struct

_WRAP = {b"\n": b"<0x0A>", b"\t": b"<0x09>", b"'": b"<0x27>", b'"': b"<0x22>"}


def write_bin(tokens, scores):
    """tokenizer_creation.py:43-48: u32 max length, then (f32 score, u32 len, bytes) per token."""
    out = [struct.pack("I", max(len(t) for t in tokens))]
    for t, s in zip(tokens, scores):
        out.append(struct.pack("fI", s, len(t)))
        out.append(t)
    return b"".join(out)


class Tokenizer:
    def __init__(self, data, vocab_size):
        """`Tokenizer.__init__` :239-251."""
        (self.max_token_length,) = struct.unpack_from("I", data, 0)
        off = 4
        self.vocab, self.scores = [], []
        for _ in range(vocab_size):
            score, n = struct.unpack_from("fI", data, off)
            off += 8
            self.vocab.append(bytes(data[off:off + n]))
            self.scores.append(score)
            off += n
        self.index = {}
        for i, t in enumerate(self.vocab):  # first id of a duplicated string
            self.index.setdefault(t, i)

    def find(self, token):
        """`find` :270-287 with `wrap` :200-209."""
        return self.index.get(_WRAP.get(token, token), -1)


def bpe_encode(text, tok):
    """`bpe_encode` :289-327 -> (ids, complete)."""
    data = text.encode() if isinstance(text, str) else text
    tokens = []
    for pos in range(len(data)):
        i = tok.find(data[pos:pos + 1])
        if i == -1:
            return tokens, False
        tokens.append(i)
    while True:
        best_score, best_id, best_idx = -1e10, -1, -1
        for i in range(len(tokens) - 1):
            j = tok.find(tok.vocab[tokens[i]] + tok.vocab[tokens[i + 1]])
            if j != -1 and tok.scores[j] > best_score:
                best_score, best_id, best_idx = tok.scores[j], j, i
        if best_idx == -1:
            break
        tokens[best_idx:best_idx + 2] = [best_id]
    return tokens, True

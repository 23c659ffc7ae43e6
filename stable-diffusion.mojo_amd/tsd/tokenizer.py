"""Host-side mirror of the reference's prompt tokenizer: `Tokenizer` / `bpe_encode` (helpers/utils.mojo:229-327)
over the `tokenizer_clip.bin` file written by tokenizer_creation.py.  Pure host logic inside libtsd (no GPU)."""
import ctypes as C

import numpy as np

from ._lib import check, lib, vp


def process_prompt(prompt):
    """pipeline.mojo:39-40: spaces become the CLIP end-of-word marker before encoding."""
    return prompt.replace(" ", "</w>")


class Tokenizer:
    """`Tokenizer(vocab_size, buf)` helpers/utils.mojo:229-287; `vocab_size` = 49408 for CLIP (pipeline.mojo:37)."""

    def __init__(self, path=None, vocab_size=49408, data=None):
        h = vp()
        if data is not None:
            buf = bytes(data)
            check(lib().tsd_tokenizer_create_from_memory(buf, len(buf), int(vocab_size), C.byref(h)))
        else:
            check(lib().tsd_tokenizer_create(str(path).encode(), int(vocab_size), C.byref(h)))
        self.h = h
        self.vocab_size = int(vocab_size)

    def find(self, token):
        t = token if isinstance(token, bytes) else token.encode()
        return int(lib().tsd_tokenizer_find(self.h, t))

    def token(self, idx):
        out = C.create_string_buffer(512)
        score = C.c_float()
        check(lib().tsd_tokenizer_token(self.h, int(idx), out, 512, C.byref(score)))
        return out.value, float(score.value)

    def bpe_encode(self, text):
        """`bpe_encode(text, tok)` helpers/utils.mojo:289-327 -> list of ids (shorter when an unknown character stops it)."""
        t = text if isinstance(text, bytes) else text.encode()
        n, done = C.c_int(), C.c_int()
        check(lib().tsd_tokenizer_encode(self.h, t, None, 0, C.byref(n), C.byref(done)))
        ids = np.zeros(max(1, n.value), dtype=np.int32)
        check(lib().tsd_tokenizer_encode(self.h, t, ids.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n), C.byref(done)))
        if not done.value:
            print("Not a good prompt token at pos ", n.value)  # helpers/utils.mojo:294
        return [int(v) for v in ids[: n.value]]

    def close(self):
        if self.h:
            lib().tsd_tokenizer_destroy(self.h)
            self.h = None

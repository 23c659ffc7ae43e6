"""Prompt tokenizer (host-only part of libtsd): product vs the oracle restatement on a synthetic
vocabulary in the tokenizer_clip.bin wire format; hand-checked merges; edge cases.  No GPU needed."""
import numpy as np
import pytest

from oracle import tokenizer as otok


def _vocab():
    singles = [bytes([c]) for c in range(32, 127) if chr(c) not in "'\""] + [b"<0x27>", b"<0x22>", b"<0x0A>", b"<0x09>"]
    merges = [b"th", b"he", b"the", b"</w>", b"</", b"w>", b"<", b"at", b"cat", b"ca", b"the</w>", b"a</w>", b"</w>c",
              b"on", b"on</w>", b"ma", b"mat", b"s</w>", b"ts", b"sat", b"sa", b"he</w>"]
    tokens, seen = [], set()
    for t in singles + merges + [b"at"]:  # one deliberate duplicate ("at" twice)
        tokens.append(t)
        seen.add(t)
    rng = np.random.RandomState(3)
    scores = [float(rng.randint(0, 50)) for _ in tokens]
    return tokens, scores


@pytest.fixture(scope="module")
def toks(tsd_mod):
    tokens, scores = _vocab()
    data = otok.write_bin(tokens, scores)
    prod = tsd_mod.Tokenizer(data=data, vocab_size=len(tokens))
    ref = otok.Tokenizer(data, len(tokens))
    yield prod, ref, tokens, scores
    prod.close()


def test_parse_matches_oracle(toks):
    prod, ref, tokens, scores = toks
    for i in (0, 5, len(tokens) - 1):
        t, s = prod.token(i)
        assert t == ref.vocab[i] == tokens[i] and s == np.float32(scores[i])
    assert prod.find(b"the") == ref.find(b"the") == tokens.index(b"the")
    assert prod.find(b"at") == tokens.index(b"at")            # first id of the duplicated entry
    assert prod.find(b"zzz") == ref.find(b"zzz") == -1
    assert prod.find(b"'") == ref.find(b"'") == tokens.index(b"<0x27>")   # wrap, helpers/utils.mojo:200-209
    assert prod.find(b"\n") == tokens.index(b"<0x0A>")


@pytest.mark.parametrize("prompt", ["the cat sat on the mat", "a cat", "cats that math", "", "t", "he said 'the' \"cat\"",
                                    "the\tcat\nsat", "<w>", "   "])
def test_bpe_encode_matches_oracle(toks, tsd_mod, prompt):
    prod, ref, tokens, _ = toks
    text = tsd_mod.process_prompt(prompt)                     # pipeline.mojo:39: " " -> "</w>"
    ids = prod.bpe_encode(text)
    want, complete = otok.bpe_encode(text, ref)
    assert complete and ids == want
    # decoding the ids gives back the processed text (quotes / control characters through their <0xNN> names)
    names = {b"<0x27>": b"'", b"<0x22>": b'"', b"<0x0A>": b"\n", b"<0x09>": b"\t"}
    assert b"".join(names.get(tokens[i], tokens[i]) for i in ids) == text.encode()


def test_bpe_merge_order_by_score(tsd_mod):
    """Hand-derived: the pair with the highest score merges first, ties keep the leftmost (helpers/utils.mojo:300-309)."""
    tokens = [b"a", b"b", b"c", b"ab", b"bc", b"abc"]

    def enc(scores):
        data = otok.write_bin(tokens, [float(s) for s in scores])
        t = tsd_mod.Tokenizer(data=data, vocab_size=6)
        got = t.bpe_encode("abc")
        t.close()
        assert got == otok.bpe_encode("abc", otok.Tokenizer(data, 6))[0]
        return got

    assert enc([0, 0, 0, 1, 2, 9]) == [5]        # "bc" (2) beats "ab" (1); then "a" + "bc" = "abc" is in the vocabulary
    assert enc([0, 0, 0, 5, 2, 9]) == [5]        # "ab" first, then "ab" + "c"
    assert enc([0, 0, 0, 3, 3, -20]) == [5]      # tie -> leftmost; any score above -1e10 still merges
    tokens = [b"a", b"b", b"c", b"ab", b"bc", b"xyz"]
    assert enc([0, 0, 0, 1, 2, 9]) == [0, 4]     # no "abc": a + bc stays two tokens
    assert enc([0, 0, 0, 2, 2, 9]) == [3, 2]     # tie between "ab" and "bc" -> leftmost pair


def test_unknown_character_stops_early(toks, capsys):
    prod, ref, _, _ = toks
    ids = prod.bpe_encode("cat\x01dog")
    assert ids == otok.bpe_encode("cat\x01dog", ref)[0] == [prod.find(b"c"), prod.find(b"a"), prod.find(b"t")]
    assert "Not a good prompt token" in capsys.readouterr().out   # helpers/utils.mojo:294


def test_truncated_file_is_an_error(tsd_mod):
    tokens, scores = _vocab()
    data = otok.write_bin(tokens, scores)
    with pytest.raises(tsd_mod.TsdError):
        tsd_mod.Tokenizer(data=data[: len(data) // 2], vocab_size=len(tokens))
    with pytest.raises(tsd_mod.TsdError):
        tsd_mod.Tokenizer(path="/nonexistent/tokenizer_clip.bin")

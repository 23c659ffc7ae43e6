// tokenizer.cpp - host-side prompt tokenizer of the reference (no GPU code): the `Tokenizer` struct and `bpe_encode`
// of helpers/utils.mojo:229-327 over the `tokenizer_clip.bin` wire format written by tokenizer_creation.py:43-48
// (u32 max_token_length, then per token: f32 score, u32 byte length, bytes).  SURVEY.md section 8 f-3.
//
// Intended semantics where the literal code is broken: `str_concat` (helpers/utils.mojo:221-231)
// copies the first byte of each operand repeatedly - the concatenation is meant; `bpe_encode` reads the score of
// id -1 before testing it (:303-305) - the test comes first here; `wrap` (:200-209) compares one character against
// the two-character literals "\\n" / "\\t" - newline and tab map to <0x0A> / <0x09> as the byte-fallback names say.
#include <stdio.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

struct tsd_tokenizer {
  std::vector<std::string> vocab;
  std::vector<float> scores;
  std::unordered_map<std::string, int> index;  // first id of each distinct string (the reference's binary search over
                                               // the sorted vocabulary returns an unspecified one among duplicates)
  int max_token_length = 0;
};

static const char* wrap_char(const std::string& s) {  // helpers/utils.mojo:200-209
  if (s == "\n") return "<0x0A>";
  if (s == "\t") return "<0x09>";
  if (s == "'") return "<0x27>";
  if (s == "\"") return "<0x22>";
  return nullptr;
}
static int tok_find(const tsd_tokenizer* t, const std::string& s) {  // Tokenizer.find :270-287
  const char* w = wrap_char(s);
  auto it = t->index.find(w ? std::string(w) : s);
  return it == t->index.end() ? -1 : it->second;
}

extern "C" int tsd_tokenizer_create_from_memory(const void* data, size_t bytes, int vocab_size, tsd_tokenizer** out) {
  if (!data || !out) TSD_FAIL(TSD_E_ARG, "tsd_tokenizer_create_from_memory: NULL argument");
  if (vocab_size <= 0) TSD_FAIL(TSD_E_ARG, "tokenizer: vocab_size %d", vocab_size);
  *out = nullptr;
  const unsigned char* p = (const unsigned char*)data;
  size_t off = 0;
  auto need = [&](size_t n) { return off + n <= bytes; };
  if (!need(4)) TSD_FAIL(TSD_E_SHAPE, "tokenizer: file too short for its header");
  tsd_tokenizer* t = new tsd_tokenizer();
  unsigned u;
  memcpy(&u, p + off, 4); off += 4;
  t->max_token_length = (int)u;  // Tokenizer.__init__ :240
  t->vocab.reserve(vocab_size);
  t->scores.reserve(vocab_size);
  for (int i = 0; i < vocab_size; i++) {  // :246-250
    float score; unsigned len;
    if (!need(8)) { delete t; TSD_FAIL(TSD_E_SHAPE, "tokenizer: file ends inside the record header of token %d of %d", i, vocab_size); }
    memcpy(&score, p + off, 4); memcpy(&len, p + off + 4, 4); off += 8;
    if (!need(len)) { delete t; TSD_FAIL(TSD_E_SHAPE, "tokenizer: file ends inside token %d (%u bytes)", i, len); }
    t->vocab.emplace_back((const char*)p + off, (size_t)len);
    t->scores.push_back(score);
    off += len;
    t->index.emplace(t->vocab.back(), i);  // keeps the first id of a duplicated string
  }
  *out = t;
  return TSD_OK;
}

extern "C" int tsd_tokenizer_create(const char* path, int vocab_size, tsd_tokenizer** out) {
  if (!path || !out) TSD_FAIL(TSD_E_ARG, "tsd_tokenizer_create: NULL argument");
  FILE* f = fopen(path, "rb");
  if (!f) TSD_FAIL(TSD_E_ARG, "tokenizer: cannot open %s", path);  // read_file :99-109 prints "Error reading file"
  std::vector<unsigned char> buf;
  unsigned char tmp[1 << 16];
  size_t n;
  while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  fclose(f);
  return tsd_tokenizer_create_from_memory(buf.data(), buf.size(), vocab_size, out);
}

extern "C" int tsd_tokenizer_destroy(tsd_tokenizer* t) {
  delete t;
  return TSD_OK;
}

extern "C" int tsd_tokenizer_find(const tsd_tokenizer* t, const char* token) {
  if (!t || !token) return -1;
  return tok_find(t, token);
}

extern "C" int tsd_tokenizer_token(const tsd_tokenizer* t, int id, char* out, int cap, float* score) {
  if (!t || id < 0 || id >= (int)t->vocab.size()) TSD_FAIL(TSD_E_ARG, "tokenizer: id %d out of range", id);
  if (out && cap > 0) {
    const size_t n = std::min((size_t)cap - 1, t->vocab[id].size());
    memcpy(out, t->vocab[id].data(), n);
    out[n] = 0;
  }
  if (score) *score = t->scores[id];
  return TSD_OK;
}

// `bpe_encode` helpers/utils.mojo:289-327: one id per character (byte), then repeatedly merge the adjacent pair whose
// concatenation is in the vocabulary with the highest score (first such pair on ties) until no pair merges.
// A character that is not in the vocabulary ends the encoding early with the ids so far (:292-296); *complete is 0 then.
extern "C" int tsd_tokenizer_encode(const tsd_tokenizer* t, const char* text, int32_t* ids, int cap, int* n_out,
                                    int* complete) {
  if (!t || !text || !n_out) TSD_FAIL(TSD_E_ARG, "tsd_tokenizer_encode: NULL argument");
  std::vector<int> tokens;
  bool ok = true;
  for (const char* c = text; *c; c++) {
    const int id = tok_find(t, std::string(1, *c));
    if (id == -1) { ok = false; break; }
    tokens.push_back(id);
  }
  if (ok) {
    for (;;) {
      float best_score = -1e10f;
      int best_id = -1, best_idx = -1;
      for (size_t i = 0; i + 1 < tokens.size(); i++) {
        const int id = tok_find(t, t->vocab[tokens[i]] + t->vocab[tokens[i + 1]]);
        if (id != -1 && t->scores[id] > best_score) { best_score = t->scores[id]; best_id = id; best_idx = (int)i; }
      }
      if (best_idx == -1) break;
      tokens[best_idx] = best_id;
      tokens.erase(tokens.begin() + best_idx + 1);
    }
  }
  *n_out = (int)tokens.size();
  if (complete) *complete = ok ? 1 : 0;
  if (ids) {
    if ((int)tokens.size() > cap) TSD_FAIL(TSD_E_SHAPE, "tokenizer: %d ids do not fit in a buffer of %d", (int)tokens.size(), cap);
    for (size_t i = 0; i < tokens.size(); i++) ids[i] = tokens[i];
  }
  return TSD_OK;
}

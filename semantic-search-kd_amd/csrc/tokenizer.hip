// Host-side BERT WordPiece tokenizer (uncased) for the text -> embedding path.
//
// The reference tokenises with the Rust `tokenizers` library behind
// SentenceTransformer.encode (reference: src/models/student.py's SentenceTransformer call sites,
// src/utils/chunk.py:26 AutoTokenizer.from_pretrained; WordPiece uncased per SURVEY.md §8c).
// Through its Python binding the per-text Encoding objects and id lists cost ~0.1 ms of
// single-threaded interpreter time per passage - 20x slower than the MI355X encodes them - so the
// product path tokenises here: plain C++ threads over the texts, ids written straight into the
// flat int32 stream `sskd_pack_tokens` consumes.
//
// Scope: exact BertNormalizer(clean_text, lowercase) + BertPreTokenizer + WordPiece +
// "[CLS] $A [SEP]" semantics for ASCII input.  A text with any byte >= 0x80 (accents to strip,
// CJK spacing, Unicode punctuation / whitespace classes) is FLAGGED, not approximated: the caller
// runs those texts through the `tokenizers` library.  Host code only - no device work here.
#include "common.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct PieceTable {
  // open addressing over (offset, length) slices of one character pool; FNV-1a
  std::vector<int32_t> slot_id;     // -1 = empty
  std::vector<uint32_t> slot_off;
  std::vector<uint16_t> slot_len;
  uint32_t mask = 0;
  int max_len = 0;

  static uint32_t hash(const char* s, int n) {
    uint32_t h = 2166136261u;
    for (int i = 0; i < n; ++i) h = (h ^ (uint8_t)s[i]) * 16777619u;
    return h;
  }
  void init(size_t n_items) {
    size_t cap = 64;
    while (cap < n_items * 3) cap <<= 1;
    slot_id.assign(cap, -1);
    slot_off.assign(cap, 0);
    slot_len.assign(cap, 0);
    mask = (uint32_t)cap - 1;
  }
  void insert(const std::string& pool, uint32_t off, int len, int id) {
    uint32_t h = hash(pool.data() + off, len) & mask;
    while (slot_id[h] >= 0) {
      if (slot_len[h] == len && memcmp(pool.data() + slot_off[h], pool.data() + off, len) == 0) return;  // first wins
      h = (h + 1) & mask;
    }
    slot_id[h] = id;
    slot_off[h] = off;
    slot_len[h] = (uint16_t)len;
    if (len > max_len) max_len = len;
  }
  int find(const std::string& pool, const char* s, int len) const {
    if (len > max_len) return -1;
    uint32_t h = hash(s, len) & mask;
    while (slot_id[h] >= 0) {
      if (slot_len[h] == len && memcmp(pool.data() + slot_off[h], s, len) == 0) return slot_id[h];
      h = (h + 1) & mask;
    }
    return -1;
  }
};

struct Tokenizer {
  std::string pool;
  PieceTable first, cont;  // word-initial pieces, "##" continuation pieces (stored without "##")
  int unk = -1, cls = -1, sep = -1;
  int max_chars_per_word = 100;
};

inline bool is_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
inline bool is_ctrl(unsigned char c) { return (c < 0x20 && !is_ws(c)) || c == 0x7f; }
inline bool is_punct(unsigned char c) {
  return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
}

// one text -> ids (without truncation); returns false when the text needs the Unicode path
bool encode_one(const Tokenizer& tk, const char* s, int64_t n, std::vector<int32_t>& out, std::string& word) {
  out.push_back(tk.cls);
  auto flush_word = [&]() {
    if (word.empty()) return;
    const int len = (int)word.size();
    if (len > tk.max_chars_per_word) {
      out.push_back(tk.unk);
      word.clear();
      return;
    }
    const size_t mark = out.size();
    int start = 0;
    bool bad = false;
    while (start < len) {
      const PieceTable& tb = start == 0 ? tk.first : tk.cont;
      int end = std::min(len, start + tb.max_len), id = -1;
      for (; end > start; --end) {
        id = tb.find(tk.pool, word.data() + start, end - start);
        if (id >= 0) break;
      }
      if (id < 0) {
        bad = true;
        break;
      }
      out.push_back(id);
      start = end;
    }
    if (bad) {
      out.resize(mark);
      out.push_back(tk.unk);
    }
    word.clear();
  };
  for (int64_t i = 0; i < n; ++i) {
    const unsigned char c = (unsigned char)s[i];
    if (c >= 0x80) return false;
    if (c == 0 || is_ctrl(c)) continue;  // BertNormalizer clean_text: dropped
    if (is_ws(c)) {
      flush_word();
    } else if (is_punct(c)) {
      flush_word();
      word.push_back((char)c);
      flush_word();
    } else {
      word.push_back((char)((c >= 'A' && c <= 'Z') ? c + 32 : c));
    }
  }
  flush_word();
  out.push_back(tk.sep);
  return true;
}

}  // namespace

extern "C" {

int sskd_tokenizer_create(const char* vocab_blob, int64_t blob_bytes, void** handle) {
  SSKD_REQUIRE(vocab_blob && blob_bytes > 0 && handle, "tokenizer_create: null / empty vocabulary");
  auto* tk = new Tokenizer();
  tk->pool.assign(vocab_blob, (size_t)blob_bytes);
  std::vector<std::pair<uint32_t, int>> lines;  // (offset, length) per id
  uint32_t off = 0;
  for (int64_t i = 0; i <= blob_bytes; ++i) {
    if (i == blob_bytes || vocab_blob[i] == '\n') {
      if (i > off || i < blob_bytes) lines.emplace_back(off, (int)(i - off));
      off = (uint32_t)i + 1;
    }
  }
  tk->first.init(lines.size());
  tk->cont.init(lines.size());
  for (int id = 0; id < (int)lines.size(); ++id) {
    const char* p = tk->pool.data() + lines[id].first;
    const int len = lines[id].second;
    if (len == 0 || len > 0xffff) continue;
    if (len == 5 && memcmp(p, "[UNK]", 5) == 0) tk->unk = id;
    if (len == 5 && memcmp(p, "[CLS]", 5) == 0) tk->cls = id;
    if (len == 5 && memcmp(p, "[SEP]", 5) == 0) tk->sep = id;
    if (len > 2 && p[0] == '#' && p[1] == '#') tk->cont.insert(tk->pool, lines[id].first + 2, len - 2, id);
    else tk->first.insert(tk->pool, lines[id].first, len, id);
  }
  if (tk->unk < 0 || tk->cls < 0 || tk->sep < 0) {
    delete tk;
    return sskd::fail(SSKD_ERR_INVALID, "tokenizer_create: vocabulary lacks [UNK] / [CLS] / [SEP]");
  }
  *handle = tk;
  return SSKD_OK;
}

void sskd_tokenizer_destroy(void* handle) { delete static_cast<Tokenizer*>(handle); }

int sskd_tokenizer_encode(void* handle, const char* text_blob, const int64_t* offsets, int n_texts,
                          int max_len, int n_threads, int32_t* out_ids, int64_t out_capacity,
                          int32_t* out_lengths, uint8_t* out_needs_unicode, int64_t* out_total) {
  SSKD_REQUIRE(handle && offsets && out_lengths && out_needs_unicode && out_total, "tokenizer_encode: null pointer");
  SSKD_REQUIRE(n_texts >= 0 && max_len >= 2, "tokenizer_encode: bad n_texts / max_len");
  const Tokenizer& tk = *static_cast<Tokenizer*>(handle);
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_texts) n_threads = n_texts > 0 ? n_texts : 1;
  std::vector<std::vector<int32_t>> parts(n_threads);
  auto work = [&](int t) {
    const int lo = (int)((int64_t)n_texts * t / n_threads), hi = (int)((int64_t)n_texts * (t + 1) / n_threads);
    std::vector<int32_t>& dst = parts[t];
    std::vector<int32_t> ids;
    std::string word;
    for (int i = lo; i < hi; ++i) {
      ids.clear();
      word.clear();
      const bool ok = encode_one(tk, text_blob + offsets[i], offsets[i + 1] - offsets[i], ids, word);
      out_needs_unicode[i] = ok ? 0 : 1;
      if (!ok) {
        out_lengths[i] = 0;
        continue;
      }
      int len = (int)ids.size();
      if (len > max_len) {  // keep the closing [SEP]
        ids[max_len - 1] = ids[len - 1];
        len = max_len;
      }
      out_lengths[i] = len;
      dst.insert(dst.end(), ids.begin(), ids.begin() + len);
    }
  };
  if (n_threads == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  int64_t total = 0;
  for (auto& p : parts) total += (int64_t)p.size();
  *out_total = total;
  if (total > out_capacity) return sskd::fail(SSKD_ERR_WORKSPACE, "tokenizer_encode: %lld ids > capacity %lld",
                                              (long long)total, (long long)out_capacity);
  SSKD_REQUIRE(total == 0 || out_ids, "tokenizer_encode: null output");
  int64_t pos = 0;
  for (auto& p : parts) {
    if (!p.empty()) memcpy(out_ids + pos, p.data(), p.size() * sizeof(int32_t));
    pos += (int64_t)p.size();
  }
  return SSKD_OK;
}

}  // extern "C"

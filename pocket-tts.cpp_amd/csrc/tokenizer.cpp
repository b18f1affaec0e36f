// Text pre-step (/root/reference/src/tokenizer.cpp:27-60) with a self-contained SentencePiece decoder.
//
// The reference links libsentencepiece (an external dependency that is absent from the reference tree
// and from this build environment as a C++ library).  The algorithm restated here is SentencePiece's
// published one: NFKC-style normalisation through the model's precompiled character map (a Darts
// double-array trie), whitespace escaping with U+2581, then either Viterbi segmentation over unigram
// log-probabilities or greedy highest-score pair merging for BPE, with runs of unknown symbols merged
// or expanded to byte pieces.  tests/test_tokenizer.py checks ids against the sentencepiece Python
// package on locally trained models.
#include "pocket_tts/tokenizer.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <limits>
#include <queue>
#include <stdexcept>
#include <unordered_map>

namespace pocket_tts {

namespace {

enum PieceType { kNormal = 1, kUnknown = 2, kControl = 3, kUserDefined = 4, kUnused = 5, kByte = 6 };
const char kSpaceSymbol[] = "\xe2\x96\x81";  // U+2581

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        int shift = 0;
        while (p < end) {
            uint8_t b = *p++;
            v |= static_cast<uint64_t>(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
            shift += 7;
            if (shift > 63) break;
        }
        ok = false;
        return 0;
    }
    Reader sub() {
        uint64_t n = varint();
        if (!ok || n > static_cast<uint64_t>(end - p)) { ok = false; return Reader{end, end}; }
        Reader r{p, p + n};
        p += n;
        return r;
    }
    void skip(int wt) {
        if (wt == 0) varint();
        else if (wt == 1) p += 8;
        else if (wt == 2) sub();
        else if (wt == 5) p += 4;
        else ok = false;
        if (p > end) ok = false;
    }
};

int utf8_len(unsigned char c) {
    if (c < 0x80) return 1;
    if ((c >> 5) == 0x6) return 2;
    if ((c >> 4) == 0xe) return 3;
    if ((c >> 3) == 0x1e) return 4;
    return 1;
}

}  // namespace

struct Tokenizer::Impl {
    struct Piece { std::string s; float score = 0.f; int type = kNormal; };
    std::vector<Piece> pieces;
    int model_type = 1;  // 1 unigram, 2 BPE
    bool byte_fallback = false, treat_ws_as_suffix = false;
    int unk_id = 0;
    bool add_dummy_prefix = true, remove_extra_ws = true, escape_ws = true;
    // precompiled character map
    std::vector<uint32_t> trie;
    std::string norm_blob;
    // derived
    std::unordered_map<std::string, int> piece_id;  // matchable pieces (normal, user defined, byte)
    float min_score = 0.f, max_score = 0.f;
    size_t max_piece_len = 0;
    int byte_id[256];

    void load(const std::string& path);
    std::string normalize(const std::string& in) const;
    std::pair<std::string, size_t> normalize_prefix(const char* s, size_t n) const;
    void encode_unigram(const std::string& norm, std::vector<std::pair<std::string, int>>& out) const;
    void encode_bpe(const std::string& norm, std::vector<std::pair<std::string, int>>& out) const;
    std::vector<int64_t> encode(const std::string& text) const;
};

void Tokenizer::Impl::load(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Failed to load tokenizer: cannot open " + path);
    std::vector<uint8_t> buf;
    uint8_t tmp[1 << 16];
    size_t got;
    while ((got = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    std::fclose(f);
    Reader r{buf.data(), buf.data() + buf.size()};
    std::string charsmap;
    while (!r.done() && r.ok) {
        uint64_t tag = r.varint();
        int field = static_cast<int>(tag >> 3), wt = static_cast<int>(tag & 7);
        if (field == 1 && wt == 2) {
            Reader s = r.sub();
            Piece pc;
            while (!s.done() && s.ok) {
                uint64_t t2 = s.varint();
                int f2 = static_cast<int>(t2 >> 3), w2 = static_cast<int>(t2 & 7);
                if (f2 == 1 && w2 == 2) { Reader b = s.sub(); pc.s.assign(reinterpret_cast<const char*>(b.p), b.end - b.p); }
                else if (f2 == 2 && w2 == 5) { if (s.end - s.p >= 4) { std::memcpy(&pc.score, s.p, 4); } s.p += 4; }
                else if (f2 == 3 && w2 == 0) pc.type = static_cast<int>(s.varint());
                else s.skip(w2);
            }
            if (!s.ok) r.ok = false;
            pieces.push_back(std::move(pc));
        } else if (field == 2 && wt == 2) {
            Reader s = r.sub();
            while (!s.done() && s.ok) {
                uint64_t t2 = s.varint();
                int f2 = static_cast<int>(t2 >> 3), w2 = static_cast<int>(t2 & 7);
                if (f2 == 3 && w2 == 0) model_type = static_cast<int>(s.varint());
                else if (f2 == 35 && w2 == 0) byte_fallback = s.varint() != 0;
                else if (f2 == 24 && w2 == 0) treat_ws_as_suffix = s.varint() != 0;
                else if (f2 == 40 && w2 == 0) unk_id = static_cast<int>(s.varint());
                else s.skip(w2);
            }
        } else if (field == 3 && wt == 2) {
            Reader s = r.sub();
            while (!s.done() && s.ok) {
                uint64_t t2 = s.varint();
                int f2 = static_cast<int>(t2 >> 3), w2 = static_cast<int>(t2 & 7);
                if (f2 == 2 && w2 == 2) { Reader b = s.sub(); charsmap.assign(reinterpret_cast<const char*>(b.p), b.end - b.p); }
                else if (f2 == 3 && w2 == 0) add_dummy_prefix = s.varint() != 0;
                else if (f2 == 4 && w2 == 0) remove_extra_ws = s.varint() != 0;
                else if (f2 == 5 && w2 == 0) escape_ws = s.varint() != 0;
                else s.skip(w2);
            }
        } else {
            r.skip(wt);
        }
    }
    if (!r.ok || pieces.empty()) throw std::runtime_error("Failed to load tokenizer: not a SentencePiece model: " + path);
    if (model_type != 1 && model_type != 2) throw std::runtime_error("Failed to load tokenizer: only unigram and BPE models are supported");
    if (charsmap.size() >= 4) {
        uint32_t tsz = 0;
        std::memcpy(&tsz, charsmap.data(), 4);
        if (tsz + 4 <= charsmap.size() && tsz % 4 == 0) {
            trie.resize(tsz / 4);
            std::memcpy(trie.data(), charsmap.data() + 4, tsz);
            norm_blob.assign(charsmap.data() + 4 + tsz, charsmap.size() - 4 - tsz);
        }
    }
    std::fill(byte_id, byte_id + 256, -1);
    min_score = std::numeric_limits<float>::max();
    max_score = -std::numeric_limits<float>::max();
    bool found_unk = false;
    for (size_t i = 0; i < pieces.size(); ++i) {
        const Piece& pc = pieces[i];
        if (pc.type == kUnknown && !found_unk) { unk_id = static_cast<int>(i); found_unk = true; }
        if (pc.type == kNormal || pc.type == kUserDefined) {
            piece_id.emplace(pc.s, static_cast<int>(i));
            max_piece_len = std::max(max_piece_len, pc.s.size());
        }
        if (pc.type == kNormal) { min_score = std::min(min_score, pc.score); max_score = std::max(max_score, pc.score); }
        if (pc.type == kByte && pc.s.size() == 6) {  // "<0xAB>"
            unsigned v = 0;
            if (std::sscanf(pc.s.c_str(), "<0x%02X>", &v) == 1 && v < 256) byte_id[v] = static_cast<int>(i);
        }
    }
    if (min_score > max_score) min_score = max_score = 0.f;
}

// Longest match of the character map at s; falls back to one UTF-8 character copied as is.
std::pair<std::string, size_t> Tokenizer::Impl::normalize_prefix(const char* s, size_t n) const {
    if (n == 0) return {std::string(), 0};
    size_t best_len = 0;
    uint32_t best_val = 0;
    if (!trie.empty()) {
        size_t node = 0;
        uint32_t unit = trie[0];
        node ^= (unit >> 10) << ((unit & 0x200u) >> 6);
        for (size_t i = 0; i < n; ++i) {
            const uint8_t c = static_cast<uint8_t>(s[i]);
            node ^= c;
            if (node >= trie.size()) break;
            unit = trie[node];
            if ((unit & 0x800000ffu) != c) break;
            node ^= (unit >> 10) << ((unit & 0x200u) >> 6);
            if (node >= trie.size()) break;
            if ((unit >> 8) & 1u) { best_len = i + 1; best_val = trie[node] & 0x7fffffffu; }
        }
    }
    if (best_len > 0 && best_val < norm_blob.size()) return {std::string(norm_blob.c_str() + best_val), best_len};
    size_t len = static_cast<size_t>(utf8_len(static_cast<unsigned char>(s[0])));
    bool valid = len <= n;
    for (size_t i = 1; valid && i < len; ++i) valid = (static_cast<unsigned char>(s[i]) & 0xc0) == 0x80;
    if (!valid) return {std::string("\xEF\xBF\xBD"), 1};
    return {std::string(s, len), len};
}

std::string Tokenizer::Impl::normalize(const std::string& in) const {
    const char* s = in.data();
    size_t n = in.size();
    if (remove_extra_ws) {
        while (n > 0) {
            auto p = normalize_prefix(s, n);
            if (p.first != " ") break;
            s += p.second; n -= p.second;
        }
    }
    std::string out;
    if (n == 0) return out;
    const std::string space = escape_ws ? kSpaceSymbol : " ";
    if (add_dummy_prefix && !treat_ws_as_suffix) out += space;
    bool prev_space = remove_extra_ws;
    while (n > 0) {
        auto p = normalize_prefix(s, n);
        const std::string& sp = p.first;
        size_t off = 0;
        while (prev_space && off < sp.size() && sp[off] == ' ') ++off;
        if (off < sp.size()) {
            for (size_t i = off; i < sp.size(); ++i) {
                if (escape_ws && sp[i] == ' ') out += kSpaceSymbol;
                else out.push_back(sp[i]);
            }
            prev_space = sp.back() == ' ';
        }
        s += p.second; n -= p.second;
        if (!remove_extra_ws) prev_space = false;
    }
    if (remove_extra_ws) {
        while (out.size() >= space.size() && out.compare(out.size() - space.size(), space.size(), space) == 0)
            out.resize(out.size() - space.size());
    }
    if (add_dummy_prefix && treat_ws_as_suffix) out += space;
    return out;
}

void Tokenizer::Impl::encode_unigram(const std::string& norm, std::vector<std::pair<std::string, int>>& out) const {
    const size_t n = norm.size();
    struct Node { int id = -1; float score = 0.f; long start = -1; };
    std::vector<Node> best(n + 1);
    const float unk_score = min_score - 10.0f;
    size_t pos = 0;
    while (pos < n) {
        const size_t mblen = std::min(static_cast<size_t>(utf8_len(static_cast<unsigned char>(norm[pos]))), n - pos);
        const float base = best[pos].score;
        bool has_single = false;
        const size_t lim = std::min(max_piece_len, n - pos);
        for (size_t len = 1; len <= lim; ++len) {
            auto it = piece_id.find(norm.substr(pos, len));
            if (it == piece_id.end()) continue;
            const Piece& pc = pieces[it->second];
            const float sc = pc.type == kUserDefined ? static_cast<float>(len) * max_score - 0.1f : pc.score;
            Node& t = best[pos + len];
            const float cand = base + sc;
            if (t.start == -1 || cand > t.score) { t.score = cand; t.start = static_cast<long>(pos); t.id = it->second; }
            if (!has_single && len == mblen) has_single = true;
        }
        if (!has_single) {
            Node& t = best[pos + mblen];
            const float cand = base + unk_score;
            if (t.start == -1 || cand > t.score) { t.score = cand; t.start = static_cast<long>(pos); t.id = unk_id; }
        }
        pos += mblen;
    }
    size_t end = n;
    std::vector<std::pair<std::string, int>> rev;
    while (end > 0) {
        const Node& nd = best[end];
        rev.emplace_back(norm.substr(static_cast<size_t>(nd.start), end - static_cast<size_t>(nd.start)), nd.id);
        end = static_cast<size_t>(nd.start);
    }
    out.assign(rev.rbegin(), rev.rend());
}

void Tokenizer::Impl::encode_bpe(const std::string& norm, std::vector<std::pair<std::string, int>>& out) const {
    struct Sym { int prev, next; std::string s; };
    std::vector<Sym> syms;
    for (size_t pos = 0; pos < norm.size();) {
        size_t len = std::min(static_cast<size_t>(utf8_len(static_cast<unsigned char>(norm[pos]))), norm.size() - pos);
        Sym sy;
        sy.s = norm.substr(pos, len);
        sy.prev = static_cast<int>(syms.size()) - 1;
        sy.next = -1;
        if (!syms.empty()) syms.back().next = static_cast<int>(syms.size());
        syms.push_back(std::move(sy));
        pos += len;
    }
    struct Pair { int left, right; float score; size_t size; };
    auto cmp = [](const Pair& a, const Pair& b) { return a.score < b.score || (a.score == b.score && a.left > b.left); };
    std::priority_queue<Pair, std::vector<Pair>, decltype(cmp)> agenda(cmp);
    auto try_add = [&](int l, int r) {
        if (l < 0 || r < 0) return;
        auto it = piece_id.find(syms[l].s + syms[r].s);
        if (it == piece_id.end()) return;
        agenda.push(Pair{l, r, pieces[it->second].score, syms[l].s.size() + syms[r].s.size()});
    };
    for (int i = 1; i < static_cast<int>(syms.size()); ++i) try_add(i - 1, i);
    while (!agenda.empty()) {
        Pair top = agenda.top();
        agenda.pop();
        if (syms[top.left].s.empty() || syms[top.right].s.empty() || syms[top.left].s.size() + syms[top.right].s.size() != top.size) continue;
        if (syms[top.left].next != top.right) continue;
        syms[top.left].s += syms[top.right].s;
        syms[top.left].next = syms[top.right].next;
        if (syms[top.right].next >= 0) syms[syms[top.right].next].prev = top.left;
        syms[top.right].s.clear();
        try_add(syms[top.left].prev, top.left);
        try_add(top.left, syms[top.left].next);
    }
    out.clear();
    for (int i = syms.empty() ? -1 : 0; i >= 0; i = syms[i].next) {
        auto it = piece_id.find(syms[i].s);
        out.emplace_back(syms[i].s, it == piece_id.end() ? unk_id : it->second);
    }
}

std::vector<int64_t> Tokenizer::Impl::encode(const std::string& text) const {
    const std::string norm = normalize(text);
    std::vector<std::pair<std::string, int>> segs;
    if (!norm.empty()) {
        if (model_type == 2) encode_bpe(norm, segs);
        else encode_unigram(norm, segs);
    }
    std::vector<int64_t> ids;
    bool prev_unk = false;
    for (auto& sg : segs) {
        const bool is_unk = sg.second == unk_id;
        if (is_unk && byte_fallback) {
            for (unsigned char c : sg.first) ids.push_back(byte_id[c] >= 0 ? byte_id[c] : unk_id);
        } else if (!(prev_unk && is_unk)) {
            ids.push_back(sg.second);
        }
        prev_unk = is_unk;
    }
    return ids;
}

Tokenizer::Tokenizer(const std::string& modelPath) : impl_(std::make_unique<Impl>()) { impl_->load(modelPath); }
Tokenizer::~Tokenizer() = default;
Tokenizer::Tokenizer(Tokenizer&&) noexcept = default;
Tokenizer& Tokenizer::operator=(Tokenizer&&) noexcept = default;

std::string Tokenizer::prepareText(const std::string& text) {
    const char* ws = " \t\n\r";
    const size_t a = text.find_first_not_of(ws);
    if (a == std::string::npos) throw std::runtime_error("Text cannot be empty");
    const size_t b = text.find_last_not_of(ws);
    std::string t = text.substr(a, b - a + 1);
    if (std::isalnum(static_cast<unsigned char>(t.back()))) t += ".";
    if (std::islower(static_cast<unsigned char>(t[0]))) t[0] = static_cast<char>(std::toupper(static_cast<unsigned char>(t[0])));
    return t;
}

std::vector<int64_t> Tokenizer::encode(const std::string& text) const { return impl_->encode(prepareText(text)); }
std::vector<int64_t> Tokenizer::encodeRaw(const std::string& text) const { return impl_->encode(text); }
int Tokenizer::vocabSize() const { return static_cast<int>(impl_->pieces.size()); }

}  // namespace pocket_tts

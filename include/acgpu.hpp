// C++17 host facade over the C ABI (acgpu.h): the reference crate's public search interface with the same names,
// argument meaning and error behaviour, so that code (and tests) written against `aho_corasick::AhoCorasick` reads
// the same here.  Header-only; link with -lacgpu.
//
//   reference                                             here
//   AhoCorasick::new(patterns)          ahocorasick.rs:228    AhoCorasick::new_(patterns)
//   AhoCorasick::builder()              :250                  AhoCorasick::builder()
//   AhoCorasickBuilder::{match_kind,start_kind,ascii_case_insensitive,kind,prefilter,dense_depth,byte_classes,build}
//                                       :2342-2616, :2171     same names
//   Input::new(h).span(..).range(..).anchored(..).earliest(..)   util/search.rs:83-640    same names
//   Match::{pattern,start,end,len,is_empty,span}          util/search.rs:825-930       same names
//   is_match / find / find_iter / find_overlapping_iter / try_*   ahocorasick.rs:311-620, :1021-1357
//   replace_all / replace_all_bytes / replace_all_with(_bytes)    :651-844, :1396-1560
//   stream_find_iter / stream_replace_all(_with)          :906, :1677-1843
//   kind / start_kind / match_kind / min_pattern_len / max_pattern_len / patterns_len / memory_usage  :1867-2027
//
// Rust panics (the infallible forms) become C++ exceptions of the same error types the try_ forms return:
// BuildError (util/error.rs:16-37) and MatchError (util/error.rs:170-204).  Iterators are materialised: the device
// returns the whole ordered match list of a call, FindIter / FindOverlappingIter walk that vector.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <istream>
#include <memory>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "acgpu.h"

namespace aho_corasick {

enum class MatchKind { Standard = ACGPU_MATCH_STANDARD, LeftmostFirst = ACGPU_MATCH_LEFTMOST_FIRST,
                       LeftmostLongest = ACGPU_MATCH_LEFTMOST_LONGEST };                  // util/search.rs:1052-1074
enum class StartKind { Both = ACGPU_START_BOTH, Unanchored = ACGPU_START_UNANCHORED,
                       Anchored = ACGPU_START_ANCHORED };                                  // util/search.rs:1133-1142
enum class AhoCorasickKind { NoncontiguousNFA = ACGPU_KIND_NONCONTIGUOUS_NFA, ContiguousNFA = ACGPU_KIND_CONTIGUOUS_NFA,
                             DFA = ACGPU_KIND_DFA };                                       // ahocorasick.rs:2627-2634
enum class Anchored { No = 0, Yes = 1 };                                                   // util/search.rs:784-792

// util/error.rs:16-37
class BuildError : public std::runtime_error {
public:
    enum class Kind { StateIDOverflow, PatternIDOverflow, PatternTooLong, Other };
    BuildError(Kind k, const std::string& msg) : std::runtime_error(msg), kind_(k) {}
    Kind kind() const { return kind_; }
private:
    Kind kind_;
};

// util/error.rs:170-204
class MatchError : public std::runtime_error {
public:
    enum class Kind { InvalidInputAnchored, InvalidInputUnanchored, UnsupportedStream, UnsupportedOverlapping,
                      UnsupportedEmpty, Other };
    MatchError(Kind k, const std::string& msg) : std::runtime_error(msg), kind_(k) {}
    Kind kind() const { return kind_; }
private:
    Kind kind_;
};

namespace detail {
[[noreturn]] inline void raise(int code) {
    const char* le = acgpu_last_error();
    std::string msg = acgpu_status_str(static_cast<acgpu_status>(code));
    if (le && *le) msg += std::string(": ") + le;
    switch (code) {
        case ACGPU_ERR_STATE_ID_OVERFLOW: throw BuildError(BuildError::Kind::StateIDOverflow, msg);
        case ACGPU_ERR_PATTERN_ID_OVERFLOW: throw BuildError(BuildError::Kind::PatternIDOverflow, msg);
        case ACGPU_ERR_PATTERN_TOO_LONG: throw BuildError(BuildError::Kind::PatternTooLong, msg);
        case ACGPU_ERR_INVALID_INPUT_ANCHORED: throw MatchError(MatchError::Kind::InvalidInputAnchored, msg);
        case ACGPU_ERR_INVALID_INPUT_UNANCHORED: throw MatchError(MatchError::Kind::InvalidInputUnanchored, msg);
        case ACGPU_ERR_UNSUPPORTED_STREAM: throw MatchError(MatchError::Kind::UnsupportedStream, msg);
        case ACGPU_ERR_UNSUPPORTED_OVERLAPPING: throw MatchError(MatchError::Kind::UnsupportedOverlapping, msg);
        case ACGPU_ERR_UNSUPPORTED_EMPTY: throw MatchError(MatchError::Kind::UnsupportedEmpty, msg);
        case ACGPU_ERR_INVALID_SPAN: throw std::out_of_range(msg);        // Input::set_span panics, util/search.rs:332-342
        case ACGPU_ERR_INVALID_ARGUMENT: throw std::invalid_argument(msg);
        case ACGPU_ERR_NOMEM: throw std::bad_alloc();
        default: throw std::runtime_error(msg);                           // HIP / device errors: fail loudly
    }
}
inline void check(int code) { if (code != ACGPU_OK) raise(code); }
}  // namespace detail

using PatternID = uint32_t;

struct Span {                                                             // util/search.rs:650-760
    size_t start = 0, end = 0;
    size_t len() const { return end - start; }
    bool is_empty() const { return start >= end; }
    bool operator==(const Span& o) const { return start == o.start && end == o.end; }
};

class Match {                                                             // util/search.rs:825-930
public:
    Match() = default;
    Match(PatternID pattern, Span span) : pattern_(pattern), span_(span) {}
    static Match must(size_t pattern, size_t start, size_t end) { return Match(PatternID(pattern), Span{start, end}); }
    PatternID pattern() const { return pattern_; }
    size_t start() const { return span_.start; }
    size_t end() const { return span_.end; }
    Span span() const { return span_; }
    Span range() const { return span_; }
    bool is_empty() const { return span_.is_empty(); }
    size_t len() const { return span_.len(); }
    bool operator==(const Match& o) const { return pattern_ == o.pattern_ && span_ == o.span_; }
    bool operator!=(const Match& o) const { return !(*this == o); }
private:
    PatternID pattern_ = 0;
    Span span_{};
};

class Input {                                                             // util/search.rs:83-640
public:
    Input(const uint8_t* haystack, size_t len) : hay_(haystack), len_(len), span_{0, len} {}
    Input(std::string_view h) : Input(reinterpret_cast<const uint8_t*>(h.data()), h.size()) {}
    Input(const std::string& h) : Input(std::string_view(h)) {}
    Input(const char* h) : Input(std::string_view(h)) {}
    Input(const std::vector<uint8_t>& h) : Input(h.data(), h.size()) {}
    // a haystack already resident in HBM (no reference counterpart)
    static Input device(const uint8_t* dev_ptr, size_t len) { Input i(dev_ptr, len); i.on_device_ = true; return i; }

    Input& span(Span s) { set_span(s); return *this; }
    Input& range(size_t start, size_t end) { set_span(Span{start, end}); return *this; }
    Input& anchored(Anchored mode) { anchored_ = mode; return *this; }
    Input& earliest(bool yes) { earliest_ = yes; return *this; }
    void set_span(Span s) {                                               // :332-342: end <= len && start <= end + 1
        if (!(s.end <= len_ && s.start <= s.end + 1)) throw std::out_of_range("invalid span for haystack");
        span_ = s;
    }
    void set_start(size_t start) { set_span(Span{start, span_.end}); }
    void set_end(size_t end) { set_span(Span{span_.start, end}); }
    const uint8_t* haystack() const { return hay_; }
    size_t haystack_len() const { return len_; }
    size_t start() const { return span_.start; }
    size_t end() const { return span_.end; }
    Span get_span() const { return span_; }
    Anchored get_anchored() const { return anchored_; }
    bool get_earliest() const { return earliest_; }
    bool is_done() const { return span_.start > span_.end; }             // :627-629
    bool is_on_device() const { return on_device_; }

    acgpu_input raw() const {
        acgpu_input in{};
        in.haystack = hay_; in.haystack_len = len_; in.span_start = span_.start; in.span_end = span_.end;
        in.anchored = anchored_ == Anchored::Yes; in.earliest = earliest_; in.haystack_on_device = on_device_;
        return in;
    }
private:
    const uint8_t* hay_;
    size_t len_;
    Span span_;
    Anchored anchored_ = Anchored::No;
    bool earliest_ = false, on_device_ = false;
};

class AhoCorasick;

// FindIter / FindOverlappingIter / StreamFindIter (ahocorasick.rs:2031-2130): forward iterators over the
// materialised, ordered match list of one call
class MatchIter {
public:
    explicit MatchIter(std::vector<Match> m) : m_(std::make_shared<std::vector<Match>>(std::move(m))) {}
    std::optional<Match> next() { if (i_ < m_->size()) return (*m_)[i_++]; return std::nullopt; }
    std::vector<Match>::const_iterator begin() const { return m_->begin(); }
    std::vector<Match>::const_iterator end() const { return m_->end(); }
    size_t count() const { return m_->size(); }
    const std::vector<Match>& collect() const { return *m_; }
private:
    std::shared_ptr<std::vector<Match>> m_;
    size_t i_ = 0;
};
using FindIter = MatchIter;
using FindOverlappingIter = MatchIter;
using StreamFindIter = MatchIter;

class AhoCorasickBuilder {                                                // ahocorasick.rs:2135-2616
public:
    AhoCorasickBuilder() { acgpu_config_init(&cfg_); }
    AhoCorasickBuilder& match_kind(MatchKind k) { cfg_.match_kind = int32_t(k); return *this; }
    AhoCorasickBuilder& start_kind(StartKind k) { cfg_.start_kind = int32_t(k); return *this; }
    AhoCorasickBuilder& ascii_case_insensitive(bool yes) { cfg_.ascii_case_insensitive = yes; return *this; }
    AhoCorasickBuilder& kind(std::optional<AhoCorasickKind> k) { cfg_.kind = k ? int32_t(*k) : ACGPU_KIND_AUTO; return *this; }
    AhoCorasickBuilder& prefilter(bool yes) { cfg_.prefilter = yes; return *this; }
    AhoCorasickBuilder& dense_depth(size_t depth) {
        cfg_.dense_depth_set = 1; cfg_.dense_depth = depth > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(depth); return *this;
    }
    AhoCorasickBuilder& byte_classes(bool yes) { cfg_.byte_classes = yes; return *this; }
    // GPU-side knobs (no reference counterpart): bytes per wavefront lane, count engine 0 auto / 1 walk / 2 hot / 3 pf
    AhoCorasickBuilder& gpu_chunk_bytes(uint32_t n) { cfg_.chunk_bytes = n; return *this; }
    AhoCorasickBuilder& gpu_engine(uint32_t e) { cfg_.engine = int32_t(e); return *this; }
    AhoCorasickBuilder& gpu_deterministic_routing(bool yes) { cfg_.deterministic_routing = yes; return *this; }   // no adaptive hints
    AhoCorasickBuilder& gpu_dfa_fill(bool yes) { cfg_.gpu_dfa_fill = yes; return *this; }   // DFA rows computed on the device

    template <class I> AhoCorasick build(const I& patterns) const;       // :2171-2207, throws BuildError
private:
    acgpu_config cfg_{};
};

class AhoCorasick {                                                       // ahocorasick.rs:177-2027
public:
    template <class I> static AhoCorasick new_(const I& patterns) { return AhoCorasickBuilder().build(patterns); }
    static AhoCorasickBuilder builder() { return AhoCorasickBuilder(); }

    // ---- infallible forms (panic in Rust, throw here) and their try_ twins (same behaviour in C++)
    bool is_match(const Input& input) const {                             // :311-316
        acgpu_input in = input.raw();
        int32_t r = 0;
        detail::check(acgpu_is_match(h_.get(), &in, &r));
        return r != 0;
    }
    std::optional<Match> try_find(const Input& input) const {             // :1021-1028
        acgpu_input in = input.raw();
        int32_t found = 0;
        acgpu_match m{};
        detail::check(acgpu_find(h_.get(), &in, &found, &m));
        if (!found) return std::nullopt;
        return Match(m.pattern, Span{size_t(m.start), size_t(m.end)});
    }
    std::optional<Match> find(const Input& input) const { return try_find(input); }                     // :404-407
    FindIter try_find_iter(const Input& input) const { return MatchIter(collect(acgpu_find_iter, input)); }                 // :1275-1282
    FindIter find_iter(const Input& input) const { return try_find_iter(input); }                       // :562-565
    FindOverlappingIter try_find_overlapping_iter(const Input& input) const {                           // :1350-1357
        return MatchIter(collect(acgpu_find_overlapping, input));
    }
    FindOverlappingIter find_overlapping_iter(const Input& input) const { return try_find_overlapping_iter(input); }  // :609-615

    // ---- replace_all family (:651-844, :1396-1560)
    template <class B> std::vector<uint8_t> try_replace_all_bytes(const Input& haystack, const std::vector<B>& replace_with) const {
        return replace(haystack, replace_with, 0);
    }
    template <class B> std::vector<uint8_t> replace_all_bytes(const Input& haystack, const std::vector<B>& replace_with) const {
        return try_replace_all_bytes(haystack, replace_with);
    }
    template <class B> std::string try_replace_all(std::string_view haystack, const std::vector<B>& replace_with) const {
        const std::vector<uint8_t> v = replace(Input(haystack), replace_with, ACGPU_REPLACE_UTF8_BOUNDARIES);
        return std::string(v.begin(), v.end());
    }
    template <class B> std::string replace_all(std::string_view haystack, const std::vector<B>& replace_with) const {
        return try_replace_all(haystack, replace_with);
    }
    // closure form, automaton.rs:530-550: replace_with(match, matched bytes, dst) -> keep going?
    void try_replace_all_with_bytes(const Input& haystack, std::vector<uint8_t>& dst,
                                    const std::function<bool(const Match&, std::string_view, std::vector<uint8_t>&)>& replace_with) const {
        const uint8_t* h = haystack.haystack();
        if (haystack.is_on_device()) throw std::invalid_argument("replace_all_with needs a host haystack");
        size_t last = 0;
        for (const Match& m : try_find_iter(Input(h, haystack.haystack_len()))) {
            dst.insert(dst.end(), h + last, h + m.start());
            last = m.end();
            if (!replace_with(m, std::string_view(reinterpret_cast<const char*>(h) + m.start(), m.len()), dst)) break;
        }
        dst.insert(dst.end(), h + last, h + haystack.haystack_len());
    }
    void replace_all_with_bytes(const Input& haystack, std::vector<uint8_t>& dst,
                                const std::function<bool(const Match&, std::string_view, std::vector<uint8_t>&)>& f) const {
        try_replace_all_with_bytes(haystack, dst, f);
    }

    // ---- stream search (:906, :1677-1843): `rdr` is read in chunks of `chunk_bytes`
    StreamFindIter try_stream_find_iter(std::istream& rdr, size_t chunk_bytes = size_t(64) << 20) const {
        std::vector<Match> all;
        stream(rdr, chunk_bytes, [&](const std::vector<uint8_t>&, const std::vector<Match>& ms) {
            all.insert(all.end(), ms.begin(), ms.end());
        });
        return MatchIter(std::move(all));
    }
    StreamFindIter stream_find_iter(std::istream& rdr, size_t chunk_bytes = size_t(64) << 20) const {
        return try_stream_find_iter(rdr, chunk_bytes);
    }
    void try_stream_replace_all_with(std::istream& rdr, std::ostream& wtr,
                                     const std::function<void(const Match&, std::string_view, std::ostream&)>& replace_with,
                                     size_t chunk_bytes = size_t(64) << 20) const {
        const size_t keep = patterns_len() && max_pattern_len() ? max_pattern_len() - 1 : 0;
        std::string pend;            // unreported tail of the stream
        size_t pend_abs = 0, reported = 0, total = 0;
        stream(rdr, chunk_bytes, [&](const std::vector<uint8_t>& chunk, const std::vector<Match>& ms) {
            pend.append(reinterpret_cast<const char*>(chunk.data()), chunk.size());
            total += chunk.size();
            for (const Match& m : ms) {
                if (m.start() > reported) wtr.write(pend.data() + (reported - pend_abs), std::streamsize(m.start() - reported));
                replace_with(m, std::string_view(pend.data() + (m.start() - pend_abs), m.len()), wtr);
                reported = m.end();
            }
            const size_t safe = std::max(reported, total > keep ? total - keep : 0);  // no later match starts before this
            if (safe > reported) { wtr.write(pend.data() + (reported - pend_abs), std::streamsize(safe - reported)); reported = safe; }
            pend.erase(0, reported - pend_abs);
            pend_abs = reported;
        });
        wtr.write(pend.data(), std::streamsize(pend.size()));
    }
    template <class B>
    void try_stream_replace_all(std::istream& rdr, std::ostream& wtr, const std::vector<B>& replace_with,
                                size_t chunk_bytes = size_t(64) << 20) const {
        if (replace_with.size() != patterns_len())
            throw std::invalid_argument("stream_replace_all requires a replacement for every pattern in the automaton");
        try_stream_replace_all_with(rdr, wtr, [&](const Match& m, std::string_view, std::ostream& w) {
            const std::string_view r(replace_with[m.pattern()]);
            w.write(r.data(), std::streamsize(r.size()));
        }, chunk_bytes);
    }
    template <class B>
    void stream_replace_all(std::istream& rdr, std::ostream& wtr, const std::vector<B>& replace_with,
                            size_t chunk_bytes = size_t(64) << 20) const {
        try_stream_replace_all(rdr, wtr, replace_with, chunk_bytes);
    }

    // ---- getters (:1867-2027)
    AhoCorasickKind kind() const { return AhoCorasickKind(acgpu_kind_of(h_.get())); }
    StartKind start_kind() const { return StartKind(acgpu_start_kind_of(h_.get())); }
    MatchKind match_kind() const { return MatchKind(acgpu_match_kind_of(h_.get())); }
    size_t min_pattern_len() const { return acgpu_min_pattern_len(h_.get()); }
    size_t max_pattern_len() const { return acgpu_max_pattern_len(h_.get()); }
    size_t patterns_len() const { return acgpu_patterns_len(h_.get()); }
    size_t memory_usage() const { return acgpu_memory_usage(h_.get()); }
    acgpu_automaton* raw() const { return h_.get(); }
    // an engine variant of this automaton (acgpu_set_variant; tests and A/B runs): before its first search
    void set_variant(const char* name, int32_t value) { detail::check(acgpu_set_variant(h_.get(), name, value)); }

private:
    friend class AhoCorasickBuilder;
    struct Free { void operator()(acgpu_automaton* a) const { acgpu_free(a); } };
    explicit AhoCorasick(acgpu_automaton* h) : h_(h, Free()) {}
    std::shared_ptr<acgpu_automaton> h_;   // Arc<dyn AcAutomaton>: cheap clones, shared immutable tables

    using ListFn = acgpu_status (*)(acgpu_automaton*, const acgpu_input*, acgpu_match*, size_t, size_t*);
    std::vector<Match> collect(ListFn fn, const Input& input) const {
        acgpu_input in = input.raw();
        std::vector<acgpu_match> buf(4096);
        size_t n = 0;
        for (;;) {
            const int rc = fn(h_.get(), &in, buf.data(), buf.size(), &n);
            if (rc == ACGPU_ERR_BUFFER_TOO_SMALL) { buf.resize(n); continue; }
            detail::check(rc);
            break;
        }
        std::vector<Match> out;
        out.reserve(n);
        for (size_t i = 0; i < n; i++) out.emplace_back(buf[i].pattern, Span{size_t(buf[i].start), size_t(buf[i].end)});
        return out;
    }
    template <class B>
    std::vector<uint8_t> replace(const Input& haystack, const std::vector<B>& replace_with, uint32_t flags) const {
        if (replace_with.size() != patterns_len())   // automaton.rs:442-447 asserts
            throw std::invalid_argument("replace_all requires a replacement for every pattern in the automaton");
        std::vector<const uint8_t*> ptrs;
        std::vector<size_t> lens;
        for (const auto& r : replace_with) {
            const std::string_view v(r);
            ptrs.push_back(reinterpret_cast<const uint8_t*>(v.data()));
            lens.push_back(v.size());
        }
        acgpu_input in = Input(haystack.haystack(), haystack.haystack_len()).raw();
        in.haystack_on_device = haystack.is_on_device();
        std::vector<uint8_t> out(haystack.haystack_len() + 64);
        size_t n = 0;
        for (;;) {
            const int rc = acgpu_replace_all(h_.get(), &in, ptrs.data(), lens.data(), lens.size(), flags, out.data(),
                                             out.size(), &n);
            if (rc == ACGPU_ERR_BUFFER_TOO_SMALL) { out.resize(n); continue; }
            detail::check(rc);
            break;
        }
        out.resize(n);
        return out;
    }
    template <class F> void stream(std::istream& rdr, size_t chunk_bytes, F&& on_chunk) const {
        acgpu_stream* s = nullptr;
        detail::check(acgpu_stream_begin(h_.get(), &s));
        std::unique_ptr<acgpu_stream, void (*)(acgpu_stream*)> guard(s, acgpu_stream_end);
        std::vector<uint8_t> chunk(chunk_bytes ? chunk_bytes : 1);
        std::vector<acgpu_match> raw;
        std::vector<Match> ms;
        for (;;) {
            rdr.read(reinterpret_cast<char*>(chunk.data()), std::streamsize(chunk.size()));
            const size_t got = size_t(rdr.gcount());
            if (got == 0) break;
            size_t n = 0;
            detail::check(acgpu_stream_feed(s, chunk.data(), got, 0, nullptr, &n));
            raw.resize(n);
            detail::check(acgpu_stream_matches(s, raw.data(), raw.size(), &n));
            ms.clear();
            for (size_t i = 0; i < n; i++) ms.emplace_back(raw[i].pattern, Span{size_t(raw[i].start), size_t(raw[i].end)});
            std::vector<uint8_t> view(chunk.begin(), chunk.begin() + std::ptrdiff_t(got));
            on_chunk(view, ms);
            if (!rdr) break;
        }
    }
};

template <class I> AhoCorasick AhoCorasickBuilder::build(const I& patterns) const {
    std::vector<const uint8_t*> ptrs;
    std::vector<size_t> lens;
    for (const auto& p : patterns) {
        const std::string_view v(p);
        ptrs.push_back(reinterpret_cast<const uint8_t*>(v.data()));
        lens.push_back(v.size());
    }
    acgpu_automaton* h = nullptr;
    detail::check(acgpu_build(&cfg_, ptrs.data(), lens.data(), lens.size(), &h));
    return AhoCorasick(h);
}

}  // namespace aho_corasick

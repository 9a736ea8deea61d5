// The reference's table-driven search tests (src/tests.rs:654-721: run_search_tests / run_stream_search_tests) replayed
// through the C++ facade (include/acgpu.hpp) on the device.  The vectors and the builder-configuration matrix are
// generated from tests/golden/reference_vectors.json + tests/refmatrix.py into _vectors.inc by tests/test_cpp_facade.py.
//   facade_test --list   : no device needed; prints the number of configurations / vectors compiled in
//   facade_test          : runs everything, exit code = number of failures (capped)
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "acgpu.hpp"

using namespace aho_corasick;

struct Vec {
    const char* name;
    std::vector<std::string> patterns;
    std::string haystack;
    std::vector<Match> matches;
};
struct Config {
    const char* id;
    int match_kind;   // MatchKind
    int api;          // 0 find_iter, 1 overlapping, 2 anchored find_iter
    int kind;         // 0 = None
    int start_kind;   // -1 = builder default
    int prefilter, byte_classes, casei;
    int dense_depth_set;
    unsigned long long dense_depth;
    std::vector<int> vectors;   // indices into VECTORS
};

static std::string H(const char* hex) {   // hex -> bytes
    std::string out;
    for (size_t i = 0; hex[i] && hex[i + 1]; i += 2) {
        auto v = [](char c) { return c <= '9' ? c - '0' : (c | 32) - 'a' + 10; };
        out.push_back(char(v(hex[i]) * 16 + v(hex[i + 1])));
    }
    return out;
}
static Match M(size_t p, size_t s, size_t e) { return Match::must(p, s, e); }

#include "_vectors.inc"   // static const std::vector<Vec> VECTORS; static const std::vector<Config> CONFIGS;

static AhoCorasick build(const Config& c, const Vec& v) {
    AhoCorasickBuilder b = AhoCorasick::builder();
    b.match_kind(MatchKind(c.match_kind)).prefilter(c.prefilter).byte_classes(c.byte_classes).ascii_case_insensitive(c.casei);
    if (c.kind) b.kind(AhoCorasickKind(c.kind));
    if (c.start_kind >= 0) b.start_kind(StartKind(c.start_kind));
    if (c.dense_depth_set) b.dense_depth(size_t(c.dense_depth));
    return b.build(v.patterns);
}

static std::string show(const std::vector<Match>& ms) {
    std::ostringstream o;
    for (const Match& m : ms) o << "(" << m.pattern() << "," << m.start() << "," << m.end() << ")";
    return o.str();
}

int main(int argc, char** argv) {
    size_t n_checks = 0;
    for (const Config& c : CONFIGS) n_checks += c.vectors.size();
    if (argc > 1 && !std::strcmp(argv[1], "--list")) {
        std::printf("%zu configurations, %zu vectors, %zu checks\n", CONFIGS.size(), VECTORS.size(), n_checks);
        return 0;
    }
    int failures = 0;
    size_t done = 0, stream_done = 0;
    for (const Config& c : CONFIGS) {
        for (int vi : c.vectors) {
            const Vec& v = VECTORS[size_t(vi)];
            try {
                AhoCorasick ac = build(c, v);
                Input in(v.haystack);
                std::vector<Match> got;
                if (c.api == 1) got = ac.find_overlapping_iter(in).collect();
                else if (c.api == 2) got = ac.find_iter(in.anchored(Anchored::Yes)).collect();
                else got = ac.find_iter(in).collect();
                if (got != v.matches) {
                    std::printf("FAIL %s / %s: got %s want %s\n", c.id, v.name, show(got).c_str(), show(v.matches).c_str());
                    failures++;
                }
                done++;
                // run_stream_search_tests (src/tests.rs:672-721): Standard kind, non-overlapping, no empty patterns,
                // through a small read buffer
                if (c.api == 0 && c.match_kind == int(MatchKind::Standard) && ac.patterns_len() && ac.min_pattern_len() > 0) {
                    std::istringstream rdr(v.haystack);
                    const std::vector<Match> sg = ac.stream_find_iter(rdr, 3).collect();
                    if (sg != v.matches) {
                        std::printf("FAIL(stream) %s / %s: got %s want %s\n", c.id, v.name, show(sg).c_str(), show(v.matches).c_str());
                        failures++;
                    }
                    stream_done++;
                }
            } catch (const std::exception& e) {
                std::printf("FAIL %s / %s: exception %s\n", c.id, v.name, e.what());
                failures++;
            }
            if (failures > 20) { std::printf("too many failures\n"); return failures; }
        }
    }
    // error behaviour of the facade (src/automaton.rs:397-423, :1067-1084; src/ahocorasick.rs:2778-2789)
    auto expect = [&](const char* what, MatchError::Kind k, auto&& fn) {
        try { fn(); std::printf("FAIL %s: no error\n", what); failures++; }
        catch (const MatchError& e) { if (e.kind() != k) { std::printf("FAIL %s: wrong kind (%s)\n", what, e.what()); failures++; } }
        catch (const std::exception& e) { std::printf("FAIL %s: %s\n", what, e.what()); failures++; }
    };
    const std::vector<std::string> pats{"append", "appendage", "app"};
    expect("overlapping on leftmost-first", MatchError::Kind::UnsupportedOverlapping, [&] {
        AhoCorasick::builder().match_kind(MatchKind::LeftmostFirst).build(pats).find_overlapping_iter("append"); });
    expect("anchored input on unanchored automaton", MatchError::Kind::InvalidInputAnchored, [&] {
        AhoCorasick::new_(pats).find(Input("append").anchored(Anchored::Yes)); });
    expect("unanchored input on anchored automaton", MatchError::Kind::InvalidInputUnanchored, [&] {
        AhoCorasick::builder().start_kind(StartKind::Anchored).build(pats).find("append"); });
    expect("stream on leftmost-longest", MatchError::Kind::UnsupportedStream, [&] {
        std::istringstream r("append"); AhoCorasick::builder().match_kind(MatchKind::LeftmostLongest).build(pats).stream_find_iter(r); });
    expect("stream with an empty pattern", MatchError::Kind::UnsupportedEmpty, [&] {
        std::istringstream r("append"); AhoCorasick::new_(std::vector<std::string>{"", "a"}).stream_find_iter(r); });
    // documented examples: src/ahocorasick.rs:163-175, :636-650; README.md:85-99
    {
        AhoCorasick ac = AhoCorasick::new_(std::vector<std::string>{"fox", "brown", "quick"});
        const std::vector<std::string> rw{"sloth", "grey", "slow"};
        if (ac.replace_all("The quick brown fox.", rw) != "The slow grey sloth.") { std::printf("FAIL replace_all doc example\n"); failures++; }
        std::istringstream rdr("The quick brown fox.");
        std::ostringstream wtr;
        ac.stream_replace_all(rdr, wtr, rw, 5);
        if (wtr.str() != "The slow grey sloth.") { std::printf("FAIL stream_replace_all doc example: %s\n", wtr.str().c_str()); failures++; }
        AhoCorasick lf = AhoCorasick::builder().match_kind(MatchKind::LeftmostFirst).build(pats);
        if (lf.replace_all("append the app to the appendage", std::vector<std::string>{"x", "y", "z"}) != "x the z to the xage") {
            std::printf("FAIL leftmost-first replace_all doc example\n"); failures++;
        }
        if (lf.kind() != AhoCorasickKind::DFA || lf.patterns_len() != 3 || lf.min_pattern_len() != 3 || lf.max_pattern_len() != 9 ||
            lf.match_kind() != MatchKind::LeftmostFirst || lf.start_kind() != StartKind::Unanchored) {
            std::printf("FAIL getters\n"); failures++;
        }
    }
    std::printf("%zu searches + %zu stream searches, %d failures\n", done, stream_done, failures);
    return failures;
}

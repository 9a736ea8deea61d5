// The DFA shallow-skip transition walk of one lane (k_tri_walk<DfaTriDev, DfaTriWalk>, dfa_tri.hip), shared with the
// host: the test hook acgpu_test_dfa_tri_host runs THIS code lane by lane on the CPU (tests/test_dfa_tri_tables.py).
#pragma once
#include "tri_common.hpp"

namespace acgpu {

// piece_walk: every lane jumps from candidate to candidate while its state is shallow (depth <= 2) and takes the
// reference's step sid = trans[sid + classes[byte]] (src/dfa.rs:218-226) byte by byte while it is deep; one gather per
// trip for all lanes that need one (the child's state id, or the transition).
struct DfaTriWalk : TriLane {
    const uint32_t* child = nullptr;    // depth-3 nodes: premultiplied DFA state ids
    const uint32_t* trans3 = nullptr;   // transition table, targets of depth <= 2 tagged kTriShallow
    const uint32_t* moff = nullptr;     // match-list offsets (dfa.rs:275-286)
    uint32_t stride2 = 0, max_match = 0, trans_words = 0;
    tri_flag deep = 0;
    uint32_t sid = 0;

    ACGPU_TRI_FN void account(uint32_t s, uint32_t idx) {
        if (s > max_match) return;
        const uint32_t o = (s >> stride2) - 2;
        note_event(s, idx, moff[o + 1] - moff[o]);
    }
    ACGPU_TRI_FN void piece_walk(uint32_t lim, uint32_t own_from, int32_t rel0) {
        pos = 0;
        for (;;) {
            tri_flag need = 0, is_child = 0, owned = 0;
            uint32_t j = 0;
            const uint32_t* addr = trans3;
            if (pos < lim) {
                if (deep) {
                    j = pos;
                    owned = pos >= own_from ? 1u : 0u;
                    uint32_t x = sid + s_inv[s_buf[pos]];
#if defined(ACGPU_GUARD) && defined(__HIP_DEVICE_COMPILE__)
                    if (x >= trans_words) { if (guard && atomicAdd(guard, 1ull) < 8) printf("k_dfa_tri: transition %u >= %u\n", x, trans_words); x = 0; }
#endif
                    addr = trans3 + x;
                    need = 1;
                } else {
                    uint32_t prj = 0, bitsw = 0, uc = 0;
                    if (shallow_jump(lim, own_from, prj, bitsw, uc, j, owned) == 1) {
                        addr = child + child_index(prj, bitsw, uc);
                        need = 1; is_child = 1;
                    }
                }
            }
            if (ACGPU_TRI_ANY(need != 0)) {
                if (need) {
                    const uint32_t v = *addr;
                    if (is_child) {
                        sid = v; deep = 1;
                        if (owned) account(sid, j);
                    } else {
                        pos++;   // (a DFA transition consumes the byte, wherever it lands)
                        if (v & kTriShallow) {
                            deep = 0;
                            if (sm && owned) {   // the state of depth <= 2 it landed on may end matches
                                const uint32_t p2 = pair_at(j);
                                if (s_mc2[p2]) note_event(0x80000000u | p2, j, s_mc2[p2]);
                            }
                        } else {
                            sid = v;
                            if (owned) account(sid, j);
                        }
                    }
                }
            }
            flush_events(rel0);
            if (!ACGPU_TRI_ANY(pos < lim)) break;
        }
        ua = na;
        ub = nb;
    }
};

}  // namespace acgpu

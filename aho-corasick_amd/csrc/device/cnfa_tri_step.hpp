// The contiguous-NFA shallow-skip walk of one lane (k_cnfa_tri, cnfa_tri.hip), shared with the host: the test hook
// acgpu_test_cnfa_tri_host runs THIS code lane by lane on the CPU (tests/test_cnfa_tri_tables.py).
#pragma once
#include "tri_common.hpp"

namespace acgpu {

enum : uint32_t { MD_SHALLOW = 0, MD_NOREC = 1, MD_REC = 2 };

// piece_walk: every lane jumps from candidate to candidate while its state is shallow and walks byte by byte (state
// record, class compare, failure link: contiguous.rs:186-247) while it is deep; one gather round per trip for all
// lanes that need one.
struct TriWalk : TriLane {
    const TriChild* child = nullptr;
    const uint32_t* repr3 = nullptr;
    uint32_t alen = 0, max_match = 0, repr_words = 0;
    // lane state
    uint32_t md = MD_SHALLOW, o = 0, head = 0, fail = 0, d0 = 0, d1 = 0;
    tri_flag hd1 = 0;
    uint32_t pend = 0;   // pend: 1 + index (in the piece) of the byte that led into state o, whose matches are still to be counted

    ACGPU_TRI_FN uint32_t word(uint32_t i) const {   // word i of the current state's record
        if (i == 0) return head;
        if (i == 1) return fail;
        if (i == 2) return d0;
        if (i == 3 && hd1) return d1;
        uint32_t x = o + i;
        ACGPU_TRI_BOUND(x, repr_words, "record word");
        return repr3[x];
    }
    static ACGPU_TRI_FN uint32_t byte_index(uint32_t w, uint32_t k4) {   // 0..3: the byte of w equal to k, 4: none
        const uint32_t x = w ^ k4;
        const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;
        return z ? uint32_t(__builtin_ctz(z)) >> 3 : 4u;
    }
    // the state just entered (its record is in hand) ends matches: contiguous.rs:581-598
    ACGPU_TRI_FN void account(uint32_t idx) {
        if (o > max_match) return;
        const uint32_t kind = head & 0xFFu;
        const uint32_t base = kind == 0xFFu ? 2 + alen : (kind == 0xFEu ? 3u : 2 + ((kind + 3) >> 2) + kind);
        const uint32_t packed = word(base);
        note_event(o, idx, (packed & (1u << 31)) ? 1u : packed);
    }
    // One step of contiguous.rs:186-247 from the record in hand, on the byte at `pos` (class k).
    ACGPU_TRI_FN void attempt(uint32_t k, tri_flag owned) {
        const uint32_t kind = head & 0xFFu;
        tri_flag found = 0;
        uint32_t target = 0;
        if (kind == 0xFEu) {
            found = k == ((head >> 8) & 0xFFu) ? 1u : 0u;
            target = d0;
        } else if (kind == 0xFFu) {
            uint32_t x = o + 2 + k;
            ACGPU_TRI_BOUND(x, repr_words, "dense transition");
            target = repr3[x];
            found = target != 1u /*FAIL*/ ? 1u : 0u;
        } else {
            const uint32_t tl = kind, cl = (tl + 3) >> 2, k4 = k * 0x01010101u;
            for (uint32_t i = 0; i < cl && !found; i++) {
                const uint32_t j = byte_index(word(2 + i), k4);
                if (j < 4 && i * 4 + j < tl) { target = word(2 + cl + i * 4 + j); found = 1; }
            }
        }
        if (found) { o = target; md = MD_NOREC; pend = owned ? pos + 1 : 0u; pos++; }
        else if (fail & kTriShallow) md = MD_SHALLOW;
        else { o = fail; md = MD_NOREC; }
    }
    // The gather of a trip and what follows from it: `need_child` lanes fetch the entry of the depth-3 node they enter
    // (base of the pair + rank of the bit among the pair's children), lanes without a record (MD_NOREC) fetch theirs.
    ACGPU_TRI_FN void gather(tri_flag need_child, tri_flag owned, uint32_t prj, uint32_t bitsw, uint32_t uc, uint32_t j) {
        uint32_t x = o;
        if (!need_child) ACGPU_TRI_BOUND(x, repr_words - 3, "state record");
        const uint32_t* addr = repr3 + x;
        if (need_child) addr = reinterpret_cast<const uint32_t*>(child + child_index(prj, bitsw, uc));
        const tri_u32x4 v = *reinterpret_cast<const tri_u32x4*>(addr);
        md = MD_REC;
        if (need_child) {
            o = v.x; head = v.y; fail = v.z; d0 = v.w; hd1 = 0;
            if (owned) account(j);
        } else {
            head = v.x; fail = v.y; d0 = v.z; d1 = v.w; hd1 = 1;
            if (pend) { account(pend - 1); pend = 0; }
        }
    }
    // lim: bytes of the piece in front of the chunk end (0..16); own_from: index of the first byte whose matches this
    // chunk owns (0..16)
    // rel0: position of the piece's byte 0 relative to the chunk's grid origin (events)
    ACGPU_TRI_FN void piece_walk(uint32_t lim, uint32_t own_from, int32_t rel0) {
        pos = 0;   // (pend == 0 here: a piece ends with every record fetched and every match counted)
        for (;;) {
            if (md == MD_REC && pos < lim) attempt(s_inv[s_buf[pos]], pos >= own_from ? 1u : 0u);
            tri_flag need_child = 0, owned = 0;
            uint32_t prj = 0, bitsw = 0, uc = 0, jc = 0;
            if (md == MD_SHALLOW && pos < lim) need_child = shallow_jump(lim, own_from, prj, bitsw, uc, jc, owned) == 1 ? 1u : 0u;
            const tri_flag need = need_child | (md == MD_NOREC ? 1u : 0u);
            if (ACGPU_TRI_ANY(need != 0)) {
                if (need) gather(need_child, owned, prj, bitsw, uc, jc);
            }
            flush_events(rel0);
            if (!ACGPU_TRI_ANY(md == MD_NOREC || pos < lim)) break;
        }
        ua = na;
        ub = nb;
    }
};

}  // namespace acgpu

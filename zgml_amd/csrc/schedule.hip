// schedule.hip — see schedule.h. Pure host code (compiled with the rest of the library).
#include "schedule.h"

#include <algorithm>
#include <map>

namespace zgml {

namespace {

inline Span span(uint16_t buf, uint64_t lo, uint64_t extent) { return {buf, lo, lo + (extent ? extent : 1)}; }

// extent (in elements) of a strided 2-D access: (n0-1)*s0 + (n1-1)*s1 + 1
inline uint64_t ext2(uint64_t n0, uint64_t s0, uint64_t n1, uint64_t s1) {
    if (n0 == 0 || n1 == 0) return 0;
    return (n0 - 1) * s0 + (n1 - 1) * s1 + 1;
}

// strided 2-D access n0 x n1 with strides s0, s1 from `off`: bounding interval plus, when one
// dimension is a dense run repeated at a larger stride, the (period, width) form
inline Span span2(uint16_t buf, uint64_t off, uint64_t n0, uint64_t s0, uint64_t n1, uint64_t s1) {
    Span sp = span(buf, off, ext2(n0, s0, n1, s1));
    if (n0 > 0 && n1 > 1 && s0 == 1 && s1 >= n0) sp.period = s1, sp.width = n0;
    else if (n1 > 0 && n0 > 1 && s1 == 1 && s0 >= n1) sp.period = s0, sp.width = n1;
    return sp;
}

OpAccess op_access(const zgml_device_op& op, const DynBound& b) {
    OpAccess a;
    switch (op.kind) {
        case ZGML_DOP_ELEMENTWISE: {
            const auto& e = op.u.elementwise;
            a.writes.push_back(span(e.dst, e.dst_offset, e.n));
            a.reads.push_back(span(e.src0, e.src0_offset, e.n));
            if (e.op == ZGML_OP_ADD || e.op == ZGML_OP_MUL) a.reads.push_back(span(e.src1, e.src1_offset, e.n));
            break;
        }
        case ZGML_DOP_FUSED_ELEMENTWISE: {
            const auto& f = op.u.fused_elementwise;
            a.writes.push_back(span(f.dst, f.dst_offset, f.n));
            a.reads.push_back(span(f.src, f.src_offset, f.n));
            for (uint32_t s = 0; s < f.n_steps; s++)
                if (f.steps[s].op == ZGML_OP_ADD || f.steps[s].op == ZGML_OP_MUL)
                    a.reads.push_back(span(f.steps[s].secondary_buf, f.steps[s].secondary_offset, f.n));
            break;
        }
        case ZGML_DOP_MATMUL: {
            const auto& m = op.u.matmul;
            const auto& g = m.geom;
            a.writes.push_back(span(m.dst, g.dst_offset, ext2(g.M, g.dst_row_stride, g.N, 1)));
            a.reads.push_back(span(m.a, g.a_offset, ext2(g.M, g.a_row_stride, g.K, g.a_col_stride)));
            a.reads.push_back(span(m.b, g.b_offset, ext2(g.K, g.b_row_stride, g.N, g.b_col_stride)));
            break;
        }
        case ZGML_DOP_QMATMUL: {
            const auto& q = op.u.qmatmul;
            const uint64_t irs = q.input_row_stride ? q.input_row_stride : q.K, drs = q.dst_row_stride ? q.dst_row_stride : q.N;
            a.writes.push_back(span2(q.dst, q.dst_offset, q.M, drs, q.N, 1));
            a.reads.push_back(span2(q.input, q.input_offset, q.M, irs, q.K, 1));
            break;
        }
        case ZGML_DOP_SOFTMAX:
        case ZGML_DOP_LAYERNORM:
        case ZGML_DOP_RMSNORM: {
            const auto& r = op.u.rmsnorm;
            a.writes.push_back(span(r.dst, r.dst_offset, (uint64_t)r.rows * r.cols));
            a.reads.push_back(span(r.src, r.src_offset, (uint64_t)r.rows * r.cols));
            break;
        }
        case ZGML_DOP_REDUCE: {
            const auto& r = op.u.reduce;
            a.writes.push_back(span(r.dst, r.dst_offset, r.n_out));
            a.reads.push_back(span(r.src, r.src_offset, (uint64_t)r.n_out * r.reduce_size));
            break;
        }
        case ZGML_DOP_REPEAT: {
            const auto& r = op.u.repeat;
            a.writes.push_back(span(r.dst, r.dst_offset, r.n));
            uint64_t ext = 1;
            for (int d = 0; d < 4; d++)
                if (r.src_ne[d]) ext += (uint64_t)(r.src_ne[d] - 1) * r.src_strides[d];
            a.reads.push_back(span(r.src, r.src_offset, std::max<uint64_t>(ext, std::min<uint64_t>(r.n, (uint64_t)r.src_ne[0] * r.src_ne[1] * r.src_ne[2] * r.src_ne[3]))));
            break;
        }
        case ZGML_DOP_SLICE_ASSIGN: {
            const auto& s = op.u.slice_assign;
            a.reads.push_back(span2(s.src, s.src_offset, s.rows, s.src_row_stride, s.cols, s.src_col_stride));
            if (b.kind == 1)
                a.writes.push_back({s.dst, b.lo, b.hi});
            else
                a.writes.push_back(span2(s.dst, s.dst_offset, s.rows, s.dst_row_stride, s.cols, s.dst_col_stride));
            break;
        }
        case ZGML_DOP_ROPE: {
            const auto& r = op.u.rope;
            a.writes.push_back(span(r.dst, r.dst_off, (uint64_t)r.seq_len * 2 * r.half_d));
            a.reads.push_back(span2(r.src, r.src_off, 2 * r.half_d, r.src_rs, r.seq_len, r.src_cs));
            a.reads.push_back(span(r.cos_sin, r.cs_off, ext2(2 * r.half_d, 1, r.seq_len, r.cs_cs)));
            break;
        }
        case ZGML_DOP_ATTENTION: {
            const auto& t = op.u.attention;
            const uint64_t skv = b.max_seq_kv;
            a.writes.push_back(span2(t.dst, t.dst_off, t.d_head, t.dst_rs, t.seq_q, t.dst_cs));
            a.reads.push_back(span2(t.q, t.q_off, t.d_head, t.q_rs, t.seq_q, t.q_cs));
            a.reads.push_back(span(t.k, t.k_off, ext2(t.d_head, t.k_rs, skv, t.k_cs)));
            a.reads.push_back(span(t.v, t.v_off, ext2(t.d_head, t.v_rs, skv, t.v_cs)));
            if (t.has_mask) a.reads.push_back(span(t.mask, t.mask_off, ext2(skv, t.mask_rs, t.seq_q, t.mask_cs)));
            break;
        }
        case ZGML_DOP_KVQ_STORE: { // the column is dynamic: the whole cache buffer counts as written
            const auto& st = op.u.kvq_store;
            a.reads.push_back(span(st.src, st.src_offset, st.d_head));
            a.writes.push_back(span(st.cache, 0, (uint64_t)st.n_cols * st.d_head / 4 + (uint64_t)st.n_cols * (st.d_head / st.block_size)));
            break;
        }
        case ZGML_DOP_ATTENTION_KVQ: {
            const auto& t = op.u.attention_kvq;
            const uint64_t cache = (uint64_t)t.n_cols * t.d_head / 4 + (uint64_t)t.n_cols * (t.d_head / t.block_size);
            a.writes.push_back(span2(t.dst, t.dst_off, t.d_head, 1, t.seq_q, t.dst_cs));
            a.reads.push_back(span2(t.q, t.q_off, t.d_head, 1, t.seq_q, t.q_cs));
            a.reads.push_back(span(t.k, 0, cache));
            a.reads.push_back(span(t.v, 0, cache));
            if (t.has_mask) a.reads.push_back(span(t.mask, t.mask_off, ext2(b.max_seq_kv, t.mask_rs, t.seq_q, t.mask_cs)));
            break;
        }
        default: break;
    }
    return a;
}

inline bool overlap(const Span& x, const Span& y) { return spans_overlap(x, y); }

} // namespace

Schedule build_schedule(const std::vector<zgml_device_op>& ops, const std::vector<uint64_t>& buffer_sizes,
                        const std::vector<uint64_t>& barriers, const std::vector<uint32_t>* seq_kv_bound) {
    Schedule s;
    const size_t n = ops.size();
    s.bounds.resize(n);
    // dynamic-field bounds. A dynamic store owns [dst_base_offset, next slab base on that buffer).
    std::map<uint16_t, std::vector<uint64_t>> slab_bases;
    for (size_t i = 0; i < n; i++)
        if (ops[i].kind == ZGML_DOP_SLICE_ASSIGN && ops[i].u.slice_assign.patch_stride)
            slab_bases[ops[i].u.slice_assign.dst].push_back(ops[i].u.slice_assign.dst_base_offset);
    for (auto& kv : slab_bases) {
        std::sort(kv.second.begin(), kv.second.end());
        kv.second.erase(std::unique(kv.second.begin(), kv.second.end()), kv.second.end());
    }
    for (size_t i = 0; i < n; i++) {
        if (ops[i].kind == ZGML_DOP_SLICE_ASSIGN && ops[i].u.slice_assign.patch_stride) {
            const auto& sa = ops[i].u.slice_assign;
            const auto& bases = slab_bases[sa.dst];
            auto it = std::upper_bound(bases.begin(), bases.end(), (uint64_t)sa.dst_base_offset);
            const uint64_t limit = it != bases.end() ? *it : buffer_sizes[sa.dst];
            s.bounds[i] = {1, sa.dst_base_offset, std::max<uint64_t>(limit, (uint64_t)sa.dst_base_offset + 1), 0};
        } else if (ops[i].kind == ZGML_DOP_ATTENTION) {
            s.bounds[i].kind = 2;
            s.bounds[i].max_seq_kv = ops[i].u.attention.seq_kv;
            if (seq_kv_bound && i < seq_kv_bound->size()) s.bounds[i].max_seq_kv = std::max(s.bounds[i].max_seq_kv, (*seq_kv_bound)[i]);
        } else if (ops[i].kind == ZGML_DOP_ATTENTION_KVQ) {
            s.bounds[i].kind = 3;
            s.bounds[i].max_seq_kv = ops[i].u.attention_kvq.seq_kv;
            if (seq_kv_bound && i < seq_kv_bound->size()) s.bounds[i].max_seq_kv = std::max(s.bounds[i].max_seq_kv, (*seq_kv_bound)[i]);
        }
    }
    s.access.resize(n);
    for (size_t i = 0; i < n; i++) s.access[i] = op_access(ops[i], s.bounds[i]);

    std::vector<uint64_t> position(n);
    for (size_t i = 0; i < n; i++) position[i] = i;
    levels_from_access(s.access, position, barriers, s.level, s.levels);
    return s;
}

// hazard DAG -> levels. Per buffer we keep the accesses seen so far; an item's level is one more
// than the deepest earlier item it conflicts with (RAW, WAW or WAR).
void levels_from_access(const std::vector<OpAccess>& access, const std::vector<uint64_t>& position,
                        const std::vector<uint64_t>& barriers, std::vector<uint32_t>& level,
                        std::vector<std::vector<uint32_t>>& levels) {
    struct Seen {
        Span sp;
        uint32_t level;
        bool write;
    };
    const size_t n = access.size();
    std::map<uint16_t, std::vector<Seen>> seen;
    level.assign(n, 0);
    uint32_t floor_level = 0, max_level = 0;
    size_t next_barrier = 0;
    for (size_t i = 0; i < n; i++) {
        while (next_barrier < barriers.size() && barriers[next_barrier] <= position[i]) { // nothing crosses a barrier
            floor_level = max_level + 1;
            next_barrier++;
        }
        uint32_t lv = floor_level;
        for (const Span& w : access[i].writes)
            for (const Seen& e : seen[w.buf])
                if (overlap(w, e.sp)) lv = std::max(lv, e.level + 1);
        for (const Span& r : access[i].reads)
            for (const Seen& e : seen[r.buf])
                if (e.write && overlap(r, e.sp)) lv = std::max(lv, e.level + 1);
        level[i] = lv;
        max_level = std::max(max_level, lv);
        for (const Span& w : access[i].writes) seen[w.buf].push_back({w, lv, true});
        for (const Span& r : access[i].reads) seen[r.buf].push_back({r, lv, false});
    }
    levels.assign((size_t)max_level + 1, {});
    for (size_t i = 0; i < n; i++) levels[level[i]].push_back((uint32_t)i);
}

bool dynamic_fields_in_bounds(const Schedule& s, const std::vector<zgml_device_op>& ops) {
    for (size_t i = 0; i < ops.size() && i < s.bounds.size(); i++) {
        const DynBound& b = s.bounds[i];
        if (b.kind == 1) {
            const auto& sa = ops[i].u.slice_assign;
            const uint64_t lo = sa.dst_offset, hi = lo + ext2(sa.rows, sa.dst_row_stride, sa.cols, sa.dst_col_stride);
            if (lo < b.lo || hi > b.hi) return false;
        } else if (b.kind == 2) {
            if (ops[i].u.attention.seq_kv > b.max_seq_kv) return false;
        } else if (b.kind == 3) {
            if (ops[i].u.attention_kvq.seq_kv > b.max_seq_kv) return false;
        }
    }
    return true;
}

} // namespace zgml

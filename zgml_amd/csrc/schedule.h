// schedule.h — dependency analysis of a DeviceProgram: buffer access spans per op, the hazard DAG
// (RAW / WAW / WAR) and its topological levels. The runtime batches the independent ops of one
// level into a single launch per kind (decode is launch-bound: Llama-2-7B lowers to ~6.9k
// DeviceOps per token, 32 heads x {rope, store, attention, row-store} per layer all independent).
//
// The reference's Metal planner derives the same legality from access spans
// (src/backend/program.zig:3771-3836 `opAccessSpans`); this is an independent formulation:
// no command-kind catalogue, just "ops at the same DAG depth commute".
//
// Dynamic fields: slice_assign.dst_offset and attention.seq_kv change per refresh. Spans for
// them are ASSUMED at compile time (a dynamic store stays inside its slab, seq_kv never exceeds
// its compile-time value) and `dynamic_fields_in_bounds` re-checks every refresh; when a refresh
// violates an assumption the runtime falls back to the serial 1:1 plan, so correctness never
// depends on the assumption.
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/zgml_hip.h"

namespace zgml {

struct Span {
    uint16_t buf;
    uint64_t lo, hi; // bounding interval [lo, hi) in f32 elements
    // strided 2-D accesses (a run of `width` elements every `period` elements starting at lo), 0 = the
    // whole interval: lets the row stores of different heads into one [d, T] buffer commute for T > 1
    uint64_t period = 0, width = 0;
};

struct OpAccess {
    std::vector<Span> reads, writes;
};

struct DynBound { // per op; kind 0 = none
    uint32_t kind = 0; // 1 = slice_assign (dst span must stay in [lo, hi)), 2 = attention (seq_kv <= max)
    uint64_t lo = 0, hi = 0;
    uint32_t max_seq_kv = 0;
};

struct Schedule {
    std::vector<OpAccess> access;
    std::vector<DynBound> bounds;
    std::vector<uint32_t> level;              // per op
    std::vector<std::vector<uint32_t>> levels; // op indices per level, ascending
};

// `barriers`: sorted op indices; ops before a barrier index never move after it and vice versa
// (used by the multi-GPU harness, whose collectives sit at those indices).
// `seq_kv_bound` (optional, one entry per op): the attention read extents come from max(op's seq_kv, bound)
// rather than from the op's current (possibly refreshed-down) value.
Schedule build_schedule(const std::vector<zgml_device_op>& ops, const std::vector<uint64_t>& buffer_sizes,
                        const std::vector<uint64_t>& barriers, const std::vector<uint32_t>* seq_kv_bound = nullptr);

// Levels of an arbitrary list of items with access lists (used for fused macro-ops). `position[i]`
// is the program-order op index the item stands at (for barrier placement); items must be sorted
// by position.
void levels_from_access(const std::vector<OpAccess>& access, const std::vector<uint64_t>& position,
                        const std::vector<uint64_t>& barriers, std::vector<uint32_t>& level,
                        std::vector<std::vector<uint32_t>>& levels);

inline bool spans_overlap(const Span& x, const Span& y) {
    if (x.buf != y.buf || !(x.lo < y.hi && y.lo < x.hi)) return false;
    if (x.period && x.period == y.period && x.width <= x.period && y.width <= y.period) {
        // both touch residues [lo % P, lo % P + width) modulo the same period: disjoint residue sets never meet
        const uint64_t P = x.period, a = x.lo % P, b = y.lo % P;
        const bool a_wraps = a + x.width > P, b_wraps = b + y.width > P;
        if (!a_wraps && !b_wraps && (a + x.width <= b || b + y.width <= a)) return false;
    }
    return true;
}

// true when every dynamic field of `ops` respects the bounds assumed by `s`
bool dynamic_fields_in_bounds(const Schedule& s, const std::vector<zgml_device_op>& ops);

} // namespace zgml

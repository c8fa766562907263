// chain.hip -- a dependent chain of decode GEMVs as the PHASES of one launch.
//
// Why.  One generated frame is a chain of ~400 weight-streaming launches, each depending on the one before it (diffusion head:
// 10 per solver step x 20 steps; LM: 5 per layer x 28 layers).  rocprof: a launch costs ~3 us + bytes / 5.8 TB/s in that chain --
// ~1.8 us of kernel boundary during which the HBM pipe is idle, plus the cold start of the successor (first loads leave only
// after its waves have launched and fetched their arguments).  PMC traffic equals the algorithmic bytes, so that fixed cost is
// the whole gap between 0.60 of the HBM roofline and the 0.78 a copy kernel reaches.
//
// How.  The workgroups of ALL ops of a chain are dispatched as ONE grid, ordered by phase.  The hardware dispatches the
// workgroups of a grid in order (round-robin over the 8 XCDs, each XCD in order), so every workgroup of phase p is resident or
// done before a workgroup of phase p+1 gets a slot on the same XCD: a consumer that waits for its producers can never keep a
// producer off the chip -- no co-residency requirement for the whole grid, no deadlock -- and producers never wait for
// consumers.  A phase-(p+1) workgroup that becomes resident while phase p drains issues its FIRST WEIGHT BATCH at once (weights
// do not depend on the producer), then waits for phase p's arrival counter, acquires, and reads its activations: the HBM pipe
// keeps streaming across what used to be a kernel boundary.  Hand-off = Guideline 16 of the CDNA guide (write-through payload,
// drained, sharded agent-scope counters, relaxed poll + one acquire); see gemv_body.h.  Every spin is bounded (5 ms, shared abort
// word): a chain that cannot make progress returns an error code instead of hanging the GPU.
//
// The per-op arithmetic is vv_gemv_body -- the same code, instantiated with CHAIN = 1 -- so a phase computes bit-identical
// results to the launch it replaces (same tile -> workgroup map, same K split over waves, same reduction order).
#include "gemv_body.h"

struct VVPhase {
    VVGemm g;
    int variant;               // index into the (prologue, epilogue, parts) table below
    unsigned wg0, n_wg;        // first workgroup of the phase inside the grid, number of workgroups
    unsigned n_units;          // units = tiles x K columns; workgroup w of the phase processes units w, w + n_wg, w + 2 n_wg, ...
    unsigned n_tiles;
};

// (prologue, epilogue, parts) triples a chain may contain, and the chain families that instantiate them (bit 0: the diffusion
// head's solver step; bit 1: the LM layer's projections): a family's kernel holds only its own variants
#define VV_CHAIN_VARIANTS(X)                                                                                   \
    X(0, VV_PRO_NONE, VV_EPI_STORE, 0, 1) X(1, VV_PRO_RMS_MOD, VV_EPI_SWIGLU, 0, 1) X(2, VV_PRO_RMS_MOD, VV_EPI_SWIGLU, 1, 1) \
    X(3, VV_PRO_NONE, VV_EPI_GATED_RESID, 0, 1) X(4, VV_PRO_NONE, VV_EPI_GATED_RESID, 2, 1)                    \
    X(5, VV_PRO_RMS_MOD, VV_EPI_CFG_DPM, 0, 1) X(6, VV_PRO_RMS_MOD, VV_EPI_CFG_DPM, 1, 1)                      \
    X(7, VV_PRO_NONE, VV_EPI_RESID, 0, 2) X(8, VV_PRO_NONE, VV_EPI_RESID, 2, 2)                                \
    X(9, VV_PRO_RMS, VV_EPI_SWIGLU, 0, 2) X(10, VV_PRO_RMS, VV_EPI_BIAS, 0, 2) X(11, VV_PRO_RMS, VV_EPI_BIAS, 1, 2)

namespace {

// Every phase runs 4-wave workgroups (the shape the single-op launcher picks for wide outputs); ops with few output tiles and a
// long K (o_proj, the down projections) get their parallelism from K columns (VVGemm::kgrid = 3, partial tensors summed by the
// consumer in a fixed order) instead of 8- or 16-wave workgroups: one kernel = one workgroup size.  <= 168 VGPRs: 3 per CU.
constexpr int CHAIN_WPB = 4;

// all units of this workgroup for one (prologue, epilogue) variant: the loop sits INSIDE the variant's switch case, so each variant
// is one loop around one inlined body (a shared loop around the switch made the widest variant spill)
template <int PRO, int EPI, int MR, int PARTS, int LOOP>
__device__ __forceinline__ void chain_units(const VVPhase* __restrict__ P, unsigned local, unsigned p, unsigned prev, unsigned* done, unsigned* err,
                                            unsigned n_phases, unsigned flags, unsigned char* __restrict__ smem) {
    const unsigned n_tiles = P->n_tiles, n_units = P->n_units, n_wg = P->n_wg;
    const unsigned kgrid = n_units / n_tiles;
    if constexpr (!LOOP) {                     // one unit per workgroup (the table says so for every phase): no loop state at all
        const unsigned ksb = local / n_tiles, tile = local - ksb * n_tiles;
        const VVGemm a = P->g;
        const VVChainSync cs{done, err, p, local, prev, 0u, 1u, n_phases, n_wg, flags, 1u};
        vv_gemv_body<1, PRO, EPI, MR, CHAIN_WPB, PARTS, 0, 1>(a.W, a.W2, a.X, a.Y, a.nw, a.T, a.N, a.K, a.ldx, a.ldy, a, tile, ksb, kgrid, smem, cs);
        return;
    }
#pragma unroll 1
    for (unsigned unit = local; unit < n_units; unit += n_wg) {
        const unsigned ksb = unit / n_tiles, tile = unit - ksb * n_tiles;
        const bool first = unit == local, last = unit + n_wg >= n_units;
        if (!first) __syncthreads();            // the previous unit's split-K partials in LDS have been consumed
        const VVPhase* Pu = P;
        asm volatile("" : "+s"(Pu));            // re-read the op's arguments per unit (scalar cache): nothing of the op is pinned across the loop
        const VVGemm a = Pu->g;
        const VVChainSync cs{done, err, p, local, first ? prev : 0u, 0u, 1u, n_phases, n_wg, flags, last ? 1u : 0u};
        vv_gemv_body<1, PRO, EPI, MR, CHAIN_WPB, PARTS, 0, 1>(a.W, a.W2, a.X, a.Y, a.nw, a.T, a.N, a.K, a.ldx, a.ldy, a, tile, ksb, kgrid, smem, cs);
    }
}

template <int MR, int FAM, int LOOP>
__global__ __launch_bounds__(CHAIN_WPB * 64, 3) void vv_chain_kernel(const VVPhase* __restrict__ ph, const unsigned short* __restrict__ wg2ph,
                                                                     unsigned* __restrict__ done, unsigned* __restrict__ err, const unsigned n_phases,
                                                                     const unsigned flags) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[vv_gemv_smem_bytes<1, VV_PRO_RMS_MOD, VV_EPI_SWIGLU, MR, CHAIN_WPB>()];
    const unsigned wg = blockIdx.x;
    const unsigned p = __builtin_amdgcn_readfirstlane((unsigned)wg2ph[wg]);
    const VVPhase* P = ph + p;
    const unsigned local = wg - P->wg0;
    const unsigned prev = p ? (P - 1)->n_wg : 0u;
    // Fewer workgroups than units: a phase occupies only part of the chip's workgroup slots, so the workgroups of the NEXT phase
    // are resident -- first weight batch in registers -- while this one streams, and start the moment its last arrival lands.
    switch (P->variant) {
#define X(V, PRO, EPI, PARTS, FAMS)                                                                                                      \
        case V: if constexpr (((FAMS) >> FAM) & 1) chain_units<PRO, EPI, MR, PARTS, LOOP>(P, local, p, prev, done, err, n_phases, flags, smem); break;
        VV_CHAIN_VARIANTS(X)
#undef X
        default: break;
    }
}

}  // namespace

extern "C" int vv_gemv_ok(const VVGemm* a);

// variant index of a decode GEMV inside a chain, or -1 (the op then stays a launch of its own)
extern "C" int vv_chain_variant(const VVGemm* g) {
    if (g->T < 1 || g->T > 2 || g->sl_n > 0 || g->ksplit > 0 || !vv_gemv_ok(g)) return -1;
    if (g->x_row_mod > 0 || g->add_rows_per_vec > 0) return -1;
    const int parts = g->n_xa > 0 ? 1 : (g->n_ya > 0 ? 2 : 0);
#define X(V, PRO, EPI, PARTS, FAMS) if (g->pro == PRO && g->epi == EPI && parts == PARTS) return V;
    VV_CHAIN_VARIANTS(X)
#undef X
    return -1;
}
// family mask of a variant (bit f set: family f's kernel can run it)
extern "C" int vv_chain_families(int variant) {
#define X(V, PRO, EPI, PARTS, FAMS) if (variant == V) return FAMS;
    VV_CHAIN_VARIANTS(X)
#undef X
    return 0;
}
extern "C" int vv_chain_phase_bytes() { return (int)sizeof(VVPhase); }
// fills one table entry (host memory); returns the phase's workgroup count
extern "C" unsigned vv_chain_fill(void* entry, const VVGemm* g, int variant, unsigned wg0, unsigned wg_cap) {
    VVPhase* P = (VVPhase*)entry;
    P->g = *g;
    P->variant = variant;
    P->n_tiles = (unsigned)((g->N + 15) / 16);
    P->n_units = P->n_tiles * (unsigned)(g->kgrid > 1 ? g->kgrid : 1);
    const unsigned per = (P->n_units + wg_cap - 1) / wg_cap;        // units per workgroup, then as few workgroups as that takes (balanced)
    P->wg0 = wg0;
    P->n_wg = (P->n_units + per - 1) / per;
    return P->n_wg;
}
extern "C" int vv_chain_launch(const void* phases_dev, const unsigned short* wg2ph_dev, unsigned* done_dev, unsigned* err_dev, int n_phases,
                               unsigned total_wgs, int rows, int family, int looped, hipStream_t s) {
    if (n_phases < 1 || total_wgs < 1 || rows < 1 || rows > 2 || family < 0 || family > 1) return -1;
    if (hipMemsetAsync(done_dev, 0, (size_t)n_phases * (256 + 32) * sizeof(unsigned), s) != hipSuccess) return -2;
    // knobs: VVHIP_CHAIN_PAUSE = poll pause in units of ~0.2 us (default 4); VVHIP_CHAIN_ACQ = 1 adds an agent-scope acquire after
    // every wait (debugging aid: the write-once discipline makes it unnecessary)
    static const unsigned flags = (unsigned)((getenv("VVHIP_CHAIN_PAUSE") ? atoi(getenv("VVHIP_CHAIN_PAUSE")) : 4) & 255) |
                                  ((getenv("VVHIP_CHAIN_ACQ") && atoi(getenv("VVHIP_CHAIN_ACQ"))) ? 256u : 0u);
    const dim3 grid(total_wgs), block(CHAIN_WPB * 64);
#define GO(F, L) hipLaunchKernelGGL((vv_chain_kernel<2, F, L>), grid, block, 0, s, (const VVPhase*)phases_dev, wg2ph_dev, done_dev, err_dev, (unsigned)n_phases, flags)
    if (family == 0) { if (looped) GO(0, 1); else GO(0, 0); }
    else { if (looped) GO(1, 1); else GO(1, 0); }
#undef GO
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// gemv_body.h -- the body of the decode GEMV (see gemv.hip for the design notes), shared by vv_gemv_kernel (one launch per op)
// and vv_chain_kernel (chain.hip: a dependent chain of ops as the phases of ONE launch).
#pragma once
#include "vv_common.h"


#ifdef VV_GEMM_TIMING
#define VV_STAMP(i) do { if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
// per-workgroup wall-clock (100 MHz, chip-wide) entry/exit stamps: the launch's occupancy timeline (tools/gemv_timeline.py)
#define VV_BSTAMP(i) do { if (a.dbg && threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 3200) a.dbg[16 + 2 * (blockIdx.y * gridDim.x + blockIdx.x) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VV_BSTAMP(i) do { } while (0)
#define VV_STAMP(i) do { } while (0)
#endif

namespace {

__device__ __forceinline__ float silu_f(float u) { return u / (1.0f + __expf(-u)); }
__device__ __forceinline__ float silu_acc(float u) { return u / (1.0f + expf(-u)); }
__device__ __forceinline__ float gelu_erf_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }

// full-wave sum, result uniform (returned from SGPRs): 4 DPP steps inside each row of 16 + 4 readlanes
__device__ __forceinline__ float wave_sum_dpp(float v) {
    int x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));  // row_half_mirror
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));  // row_mirror
    x = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
    return (r0 + r1) + (r2 + r3);
}

template <int XS>
__device__ __forceinline__ void split4(const float (&v)[4], uint2 (&out)[XS]) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = (__bf16)v[j];
        if constexpr (XS > 1) {
            const float r = v[j] - (float)h[j];
            m[j] = (__bf16)r;
            if constexpr (XS > 2) l[j] = (__bf16)(r - (float)m[j]);
        }
    }
    out[0] = __builtin_bit_cast(uint2, h);
    if constexpr (XS > 1) out[1] = __builtin_bit_cast(uint2, m);
    if constexpr (XS > 2) out[2] = __builtin_bit_cast(uint2, l);
}

constexpr int U = 8;       // k-steps per batch (256 k = one float4 per lane per row)

// PRO / EPI are compile-time: a launch executes only the code of its own prologue/epilogue (the runtime-
// switched version spent a third of a small launch fetching and skipping code it never needed).
// MR = activation rows a launch can carry: 4 for decode steps, 16 for prefill chunks / batched adaLN / the T = 8 codec
// stage (same weight stream, 4x the staging work and LDS).
// WPB = waves per workgroup = K split inside the workgroup.  The launcher picks it so that EVERY workgroup of the launch
// is resident at once (a second dispatch round costs a full load-latency chain): 4 for wide outputs (> 256 tiles),
// 16 for few tiles x long K (the streaming rate of a CU is set by its waves' loads in flight), 8 otherwise.
// PARTS: the activation (prologue side) or residual (epilogue side) tensor arrives as base + 2 part tensors (the producer
// split K over 3 workgroup columns): the three loads are issued together and summed in a fixed order.
// SL: slot-batched rows (VVGemm::sl_*): the rows of one launch are gathered from / scattered to the streaming buffers of up
// to 8 utterances -- one weight pass for the tokenizer stages of a whole batch.  16-row form only.

// ---- chained launch (chain.hip): a workgroup of phase p waits for every workgroup of phase p-1, then publishes itself -------
// Protocol = /opt/skills/guides/cdna_hip_programming.md, Guideline 16 (R1): payload stored WRITE-THROUGH (sc1), every storing
// wave drains (s_waitcnt vmcnt(0)), ONE lane bumps an agent-scope counter; the consumer polls relaxed agent-scope loads with
// s_sleep, does ONE agent-scope acquire (buffer_inv sc1) after the match, a workgroup barrier, then plain loads.  Counters are
// sharded 8 ways (one 128-B line each) so that the ~256 workgroups that finish together do not serialise on one word; every
// spin is bounded by the 100 MHz wall clock and a shared abort word (a launch that cannot make progress ends with an error code
// instead of hanging the GPU).
struct VVChainSync {
    unsigned* done;        // [n_phases][8 shards][32 words] arrivals per shard, then [n_phases][32 words]: shards complete ("ready")
    unsigned* err;         // abort word: non-zero once any workgroup gave up
    unsigned phase;        // this workgroup's phase
    unsigned local;        // this unit's index inside the phase
    unsigned prev_wgs;     // units (tile x K column) of phase - 1 (0: nothing to wait for)
    unsigned wave0;        // first wave of this unit inside its workgroup (0: a unit is a workgroup)
    unsigned live;         // 0: a padding unit: takes part in the barriers, loads / computes / stores nothing
    unsigned n_phases;
    unsigned my_wgs;       // units of this phase
    unsigned flags;        // bits 0..7: poll pause in units of s_sleep 8 (~0.2 us); bit 8: agent-scope acquire after the wait (not needed
                           // under the write-once discipline: every tensor a phase reads was written to lines no cache has seen in this launch)
    unsigned pub;          // 1: this call publishes the workgroup's arrival (the last unit a workgroup processes)
};
constexpr unsigned VV_CHAIN_TIMEOUT_TICKS = 5u * 1000u * 100u;        // 5 ms of the 100 MHz clock

// ONE lane polls ONE word per phase: the last arriver of each of the 8 shards bumps the phase's ready word, a consumer waits for
// it to reach the number of non-empty shards.  (Polling the 8 shard counters from every waiting workgroup put ~8 coherent loads
// per poll and workgroup on the fabric: with ~700 workgroups waiting that traffic alone slowed the weight stream.)
__device__ __forceinline__ void vv_chain_wait(const VVChainSync& cs) {
    if (cs.prev_wgs != 0u) {
        if (threadIdx.x < 64u) {
            unsigned* c = cs.done + (size_t)cs.n_phases * 256u + (size_t)(cs.phase - 1u) * 32u;
            const unsigned want = cs.prev_wgs < 8u ? cs.prev_wgs : 8u;
            const unsigned pause = cs.flags & 255u;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= want) break;
                for (unsigned i = 0; i < pause; ++i) __builtin_amdgcn_s_sleep(8);
                const bool late = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0) > VV_CHAIN_TIMEOUT_TICKS;
                const unsigned dead = late ? 0u : __hip_atomic_load(cs.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (late || dead != 0u) {
                    if (late && threadIdx.x == 0u) __hip_atomic_store(cs.err, 1u + cs.phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            if (cs.flags & 256u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void vv_chain_store4(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void vv_chain_store1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// called by the ONE storing wave of the workgroup (wave 0), after its stores; lane 0 is always among the live lanes
__device__ __forceinline__ void vv_chain_publish(const VVChainSync& cs) {
    if (!cs.pub) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63u) == 0u) {
        const unsigned sh = cs.local & 7u;
        const unsigned want = cs.my_wgs / 8u + ((sh < (cs.my_wgs & 7u)) ? 1u : 0u);
        const unsigned old = __hip_atomic_fetch_add(cs.done + (size_t)cs.phase * 256u + sh * 32u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == want)
            __hip_atomic_fetch_add(cs.done + (size_t)cs.n_phases * 256u + (size_t)cs.phase * 32u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// LDS bytes of one instantiation: staging tiles + split-K partials + sum(x^2) rows
template <int XS, int PRO, int EPI, int MR, int WPB>
constexpr int vv_gemv_smem_bytes() {
    constexpr int NM = (EPI == VV_EPI_SWIGLU) ? 2 : 1;
    constexpr int NOP = (PRO == VV_PRO_RMS_MOD) ? 2 : 1;
    constexpr int GSB = MR * 16 + (MR == 16 ? 16 : 0);
    return WPB * NOP * XS * 8 * 4 * GSB + WPB * NM * NOP * 64 * 16 + ((PRO == VV_PRO_RMS || PRO == VV_PRO_RMS_MOD) ? WPB * MR * 4 : 0);
}

template <int XS, int PRO, int EPI, int MR, int WPB, int PARTS, int SL, int CHAIN>      // PARTS: 0 none, 1 activation side, 2 residual side
__device__ __forceinline__ void vv_gemv_body(const u32x4* __restrict__ pW, const u32x4* __restrict__ pW2,
                                             const float* __restrict__ pX, float* __restrict__ pY,
                                             const float* __restrict__ pnw, int pT, int pN, int pK, int pldx,
                                             int pldy, const VVGemm& a, const unsigned tile_, const unsigned by_, const unsigned gdy_,
                                             unsigned char* __restrict__ smem, const VVChainSync cs) {
    constexpr bool DUAL = (EPI == VV_EPI_SWIGLU);
    constexpr int NM = DUAL ? 2 : 1;
    // LDS: [wave][XS][U][4][MR] x 16 B staging tiles, then [wave][NM][64] f32x4 partials, then [wave][MR] ssq
    // adaLN-modulated norm: y = rs * W.(x*nw*(1+scale)) + W.shift -- two B operands and two accumulator sets, so the
    // 1/rms of the row is only needed in the epilogue (as for plain RMSNorm) and no pre-pass over x exists
    constexpr int NOP = (PRO == VV_PRO_RMS_MOD) ? 2 : 1;
    // one (k-step, k-group) plane of the staging tile = MR rows x 16 B; the 16-row form pads it by 16 B so that the planes
    // one staging store touches fall into different bank groups (unpadded: 256-B stride = the same 4 banks, 16-way conflict)
    constexpr int GSB = MR * 16 + (MR == 16 ? 16 : 0);
    constexpr int LW = WPB;
    unsigned char* const stg_all = smem;                                                   // [LW * NOP * XS * U * 4 * GSB]
    f32x4 (*const red)[NM * NOP][64] = reinterpret_cast<f32x4 (*)[NM * NOP][64]>(smem + LW * NOP * XS * U * 4 * GSB);      // [LW][NM * NOP][64]
    float (*const ssq_sh)[MR] = reinterpret_cast<float (*)[MR]>(smem + LW * NOP * XS * U * 4 * GSB + LW * NM * NOP * 64 * 16);   // [LW][MR]
    if constexpr (!CHAIN) {
        // Pull every kernel argument into SGPRs with ONE batch of s_loads: left alone the compiler fetches
        // them lazily behind branches, i.e. 3-4 dependent ~600-cycle round trips on a launch's critical path.
        asm volatile("" ::"s"(a.mod_scale), "s"(a.mod_shift), "s"(a.addvec), "s"(a.bias), "s"(a.nscale), "s"(a.gate));
        asm volatile("" ::"s"(a.ld_mod), "s"(a.ld_gate), "s"(a.x_row_mod), "s"(a.add_rows_per_vec),
                     "s"(a.eps), "s"(a.z), "s"(a.x0p), "s"(a.coef), "s"(a.cfg), "s"(a.n_cfg));
        asm volatile("" ::"s"(a.yparts), "s"(a.xa), "s"(a.ya), "s"(a.n_xa), "s"(a.n_ya), "s"(a.part_stride));
        if constexpr (EPI == VV_EPI_CFG_DPM) asm volatile("" ::"s"(a.sde_noise));
    }
    VV_STAMP(0);
    VV_BSTAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave_g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // index in the workgroup (LDS regions)
    const int w0 = CHAIN ? (int)cs.wave0 : 0;                                          // chained launch: first wave of this unit (0 or 4)
    const int wave = wave_g - w0;                                                      // index in the unit (K split, epilogue)
    // 16-row form: grid.y walks 16-row tiles of a tall activation (tokenizer stages, prefill chunks)
    const int t_base = (MR == 16) ? (int)by_ * 16 : 0;
    const int T = min(MR, pT - t_base);
    const unsigned tile = tile_;
    const unsigned k_tiles = (unsigned)(pK + 31) >> 5;
    // K range of this workgroup (grid.y splits K for few-tile x long-K shapes so that all CUs stream), then of this wave
    const unsigned KSB = (MR <= 4) ? gdy_ : 1u;
    const unsigned ksb = (MR <= 4) ? by_ : 0u;
    const unsigned kchunk = (k_tiles + KSB - 1) / KSB;
    const unsigned kb0 = min(k_tiles, ksb * kchunk), kb1 = min(k_tiles, kb0 + kchunk);
    const unsigned kper = (kb1 - kb0 + WPB - 1) / WPB;
    const unsigned kt0 = kb0 + wave * kper;
    const unsigned kt1 = min(kb1, kt0 + kper);
    const bool has_k = kt0 < kt1 && (!CHAIN || cs.live != 0u);
    const int frow = lane & 15, fq = lane >> 4;
    unsigned char* stg = stg_all + (size_t)wave_g * (NOP * XS * U * 4 * GSB);
    const unsigned kk = lane * 4;
    const unsigned st_off = ((kk >> 5) * 4 + ((kk & 31) >> 3)) * GSB + (kk & 7) * 2;

    const u32x4* wbase = pW + (size_t)tile * k_tiles * 64 + lane;
    const u32x4* wbase2 = DUAL ? pW2 + (size_t)tile * k_tiles * 64 + lane : nullptr;

    // slot-batched rows: per-row activation offsets (wave-uniform), computed once
    unsigned xoff_sl[SL ? MR : 1];
    if constexpr (SL) {
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            const int rg = t_base + r;
            const int j = rg / a.sl_T, tt = rg - j * a.sl_T;
            xoff_sl[r] = (r < T) ? (a.sl_x ? (unsigned)(vv_slot_id(a.sl_id, j) * a.sl_x + tt * pldx) : (unsigned)(rg * pldx)) : 0u;
        }
    }
    constexpr int MODR = (PRO == VV_PRO_RMS_MOD) ? MR : 1;
    constexpr int ADDR = (PRO == VV_PRO_ADD_SILU) ? MR : 1;
    constexpr int PR = (PARTS == 1) ? MR : 1;
    struct XR { float4 x[MR]; float4 p0[PR]; float4 p1[PR]; float4 sc[MODR]; float4 sh[MODR]; float4 addv[ADDR]; float4 nwv; };
    auto x_load = [&](unsigned ktb, XR& R) {
        unsigned k = ktb * 32 + kk;
        const bool kin = k < min(kt1 * 32, (unsigned)pK);
        if (!kin) k = 0;                                   // clamped: always a legal address, masked later
        R.nwv = pnw ? *reinterpret_cast<const float4*>(pnw + k) : float4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            if (r < T) {
                const int rg = t_base + r;
                const int xr_idx = a.x_row_mod > 0 ? rg % a.x_row_mod : rg;
                if constexpr (SL) R.x[r] = *reinterpret_cast<const float4*>(pX + xoff_sl[r] + k);
                else R.x[r] = *reinterpret_cast<const float4*>(pX + (unsigned)(xr_idx * pldx) + k);
                if constexpr (PARTS == 1) {
                    R.p0[r] = *reinterpret_cast<const float4*>(a.xa + (unsigned)(xr_idx * pldx) + k);
                    R.p1[r] = *reinterpret_cast<const float4*>(a.xa + (unsigned)(a.part_stride + xr_idx * pldx) + k);
                }
                if constexpr (PRO == VV_PRO_ADD_SILU) {
                    const int av = a.add_rows_per_vec > 0 ? rg / a.add_rows_per_vec : 0;
                    R.addv[r] = *reinterpret_cast<const float4*>(a.addvec + (unsigned)(av * pK) + k);
                }
                if constexpr (PRO == VV_PRO_RMS_MOD) {
                    R.sc[r] = *reinterpret_cast<const float4*>(a.mod_scale + (unsigned)(rg * a.ld_mod) + k);
                    R.sh[r] = *reinterpret_cast<const float4*>(a.mod_shift + (unsigned)(rg * a.ld_mod) + k);
                }
            }
        }
    };
    auto w_load = [&](unsigned ktb, u32x4 (&dst)[U][NM]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned kt = min(ktb + u, kt1 - 1);      // clamped: tail k-steps re-read the last tile, MFMA skipped
            dst[u][0] = __builtin_nontemporal_load(wbase + kt * 64);
            if constexpr (DUAL) dst[u][1] = __builtin_nontemporal_load(wbase2 + kt * 64);
        }
    };
    // ---- first activation batch and first weight batch go out before anything else ----
    XR R;
    u32x4 wA[U][NM], wB[U][NM];
    if constexpr (CHAIN) {
        // a phase of a chained launch (chain.hip): the weights do not depend on the producer phase -- their first batch goes out
        // NOW and lands while this workgroup waits for the phase before it; activations, residual and gate operands only after
        // the acquire
        if (has_k) w_load(kt0, wA);
        __builtin_amdgcn_sched_barrier(0);
        vv_chain_wait(cs);
        if (has_k) x_load(kt0, R);
    } else {
        if (has_k) { x_load(kt0, R); w_load(kt0, wA); }      // x first: its wait must not drag the weight stream along
    }
    __builtin_amdgcn_sched_barrier(0);
    VV_STAMP(1);

    // ---- epilogue operands: requested now, consumed ~one weight stream later (wave 0 only) ----
    const int n0 = tile * 16 + fq * 4;
    const bool epi_lane = (wave == 0) && frow < T && n0 < pN && (!CHAIN || cs.live != 0u);
    float4 pre_y = {0.f, 0.f, 0.f, 0.f}, pre_b = {0.f, 0.f, 0.f, 0.f}, pre_g = {1.f, 1.f, 1.f, 1.f};
    float4 pre_y0 = {0.f, 0.f, 0.f, 0.f}, pre_y1 = {0.f, 0.f, 0.f, 0.f};
    unsigned yrow_off = (unsigned)((t_base + frow) * pldy);
    if constexpr (SL) {
        const int rg = min(t_base + frow, pT - 1);
        const int j = rg / a.sl_T, tt = rg - j * a.sl_T;
        yrow_off = a.sl_y ? (unsigned)(vv_slot_id(a.sl_id, j) * a.sl_y + tt * pldy) : (unsigned)(rg * pldy);
    }
    if (epi_lane) {            // N % 4 == 0 and 16-B aligned operands are launch preconditions (vv_gemv_ok)
        if constexpr (EPI == VV_EPI_BIAS || EPI == VV_EPI_BIAS_GELU || EPI == VV_EPI_RESID) {
            if (a.bias) pre_b = *reinterpret_cast<const float4*>(a.bias + n0);
        }
        if constexpr (EPI == VV_EPI_RESID || EPI == VV_EPI_GATED_RESID) {
            if constexpr (CHAIN) pre_y = *reinterpret_cast<const float4*>((a.R ? a.R : pY) + (yrow_off + (unsigned)n0));
            else pre_y = *reinterpret_cast<const float4*>(pY + (yrow_off + (unsigned)n0));
            if constexpr (PARTS == 2) {
                const float* yp0 = a.ya + (unsigned)((t_base + frow) * pldy + n0);
                pre_y0 = *reinterpret_cast<const float4*>(yp0);
                pre_y1 = *reinterpret_cast<const float4*>(yp0 + a.part_stride);
            }
            if constexpr (EPI == VV_EPI_GATED_RESID) pre_g = *reinterpret_cast<const float4*>(a.gate + (unsigned)((t_base + frow) * a.ld_gate + n0));
            else if (a.nscale) pre_g = *reinterpret_cast<const float4*>(a.nscale + n0);
        }
    }

    f32x4 acc[NM * NOP];            // [NM, 2NM): the shift operand's products (RMS_MOD)
#pragma unroll
    for (int i = 0; i < NM * NOP; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ssq[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) ssq[r] = 0.f;

    auto x_stage = [&](unsigned ktb, const XR& R) {
        const unsigned k = ktb * 32 + kk;
        const float msk = (k < min(kt1 * 32, (unsigned)pK)) ? 1.f : 0.f;
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            if (r < T) {
                float v[4] = {R.x[r].x, R.x[r].y, R.x[r].z, R.x[r].w};
                if constexpr (PARTS == 1) {
                    v[0] = (v[0] + R.p0[r].x) + R.p1[r].x; v[1] = (v[1] + R.p0[r].y) + R.p1[r].y;
                    v[2] = (v[2] + R.p0[r].z) + R.p1[r].z; v[3] = (v[3] + R.p0[r].w) + R.p1[r].w;
                }
                v[0] *= msk; v[1] *= msk; v[2] *= msk; v[3] *= msk;
                if constexpr (PRO == VV_PRO_RMS) {
                    ssq[r] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                    v[0] *= R.nwv.x; v[1] *= R.nwv.y; v[2] *= R.nwv.z; v[3] *= R.nwv.w;
                } else if constexpr (PRO == VV_PRO_RMS_MOD) {
                    ssq[r] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                    v[0] = (v[0] * R.nwv.x) * (1.f + R.sc[r].x); v[1] = (v[1] * R.nwv.y) * (1.f + R.sc[r].y);
                    v[2] = (v[2] * R.nwv.z) * (1.f + R.sc[r].z); v[3] = (v[3] * R.nwv.w) * (1.f + R.sc[r].w);
                    float sh4[4] = {R.sh[r].x * msk, R.sh[r].y * msk, R.sh[r].z * msk, R.sh[r].w * msk};
                    uint2 sparts[XS];
                    split4<XS>(sh4, sparts);
#pragma unroll
                    for (int p = 0; p < XS; ++p)
                        *reinterpret_cast<uint2*>(stg + (XS + p) * (U * 4 * GSB) + st_off + r * 16) = sparts[p];
                } else if constexpr (PRO == VV_PRO_ADD_SILU) {
                    v[0] = silu_acc(v[0] + R.addv[r].x) * msk; v[1] = silu_acc(v[1] + R.addv[r].y) * msk;
                    v[2] = silu_acc(v[2] + R.addv[r].z) * msk; v[3] = silu_acc(v[3] + R.addv[r].w) * msk;
                }
                uint2 parts[XS];
                split4<XS>(v, parts);
#pragma unroll
                for (int p = 0; p < XS; ++p)
                    *reinterpret_cast<uint2*>(stg + p * (U * 4 * GSB) + st_off + r * 16) = parts[p];
            }
        }
    };
    auto mma = [&](unsigned ktb, const u32x4 (&wb)[U][NM]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ktb + u < kt1) {
#pragma unroll
                for (int p = 0; p < XS; ++p) {
                    u32x4 f = u32x4{0u, 0u, 0u, 0u};
                    if (frow < MR) f = *reinterpret_cast<const u32x4*>(stg + (size_t)((p * U + u) * 4 + fq) * GSB + frow * 16);
                    const bf16x8 xb = __builtin_bit_cast(bf16x8, f);
#pragma unroll
                    for (int i = 0; i < NM; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[u][i]), xb, acc[i], 0, 0, 0);
                    if constexpr (NOP == 2) {
                        u32x4 f2 = u32x4{0u, 0u, 0u, 0u};
                        if (frow < MR) f2 = *reinterpret_cast<const u32x4*>(stg + (size_t)(((XS + p) * U + u) * 4 + fq) * GSB + frow * 16);
                        const bf16x8 sb = __builtin_bit_cast(bf16x8, f2);
#pragma unroll
                        for (int i = 0; i < NM; ++i)
                            acc[NM + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[u][i]), sb, acc[NM + i], 0, 0, 0);
                    }
                }
            }
        }
    };

    if (has_k) {
        x_stage(kt0, R);
        VV_STAMP(2);
        // two batches per trip, ping-ponging the weight buffers: no register copies, so the prefetched batch
        // stays in flight across the MFMAs of the current one
        if constexpr (DUAL) {
            // two weight streams: a second weight buffer costs 64 VGPRs and halves the resident workgroups per CU
            // (1 instead of 2); run single-buffered and let the other resident waves cover the load latency
#pragma unroll 1
            for (unsigned ktb = kt0; ktb < kt1; ktb += U) {
                const bool n1 = ktb + U < kt1;
                if (n1) x_load(ktb + U, R);
                mma(ktb, wA);
                if (n1) { w_load(ktb + U, wA); x_stage(ktb + U, R); }
            }
        } else
#pragma unroll 1
        for (unsigned ktb = kt0; ktb < kt1; ktb += 2 * U) {
            const bool n1 = ktb + U < kt1, n2 = ktb + 2 * U < kt1;
            if (n1) { x_load(ktb + U, R); w_load(ktb + U, wB); }
            mma(ktb, wA);
            if (n1) x_stage(ktb + U, R);
            if (n2) { x_load(ktb + 2 * U, R); w_load(ktb + 2 * U, wA); }
            if (n1) mma(ktb + U, wB);
            if (n2) x_stage(ktb + 2 * U, R);
        }
    }
    // rows beyond T were never staged: their fragment slots hold stale LDS -> D columns >= T are garbage, never stored.

    VV_STAMP(3);
    // ---- split-K partials -> LDS, one barrier, wave 0 finishes ----
#pragma unroll
    for (int i = 0; i < NM * NOP; ++i) red[wave_g][i][lane] = acc[i];
    if constexpr (PRO == VV_PRO_RMS || PRO == VV_PRO_RMS_MOD) {
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            const float s = (r < T) ? wave_sum_dpp(ssq[r]) : 0.f;
            if (lane == 0) ssq_sh[wave_g][r] = s;
        }
    }
    VV_STAMP(4);
    __syncthreads();
    VV_STAMP(5);
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < WPB; ++w)
#pragma unroll
        for (int i = 0; i < NM * NOP; ++i) acc[i] += red[w0 + w][i][lane];
    if (!epi_lane) return;
    float rs = 1.0f;
    if constexpr (PRO == VV_PRO_RMS || PRO == VV_PRO_RMS_MOD) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) s += ssq_sh[w0 + w][frow];
        rs = rsqrtf(s / (float)pK + a.eps);
    }
    float o[4] = {acc[0][0] * rs, acc[0][1] * rs, acc[0][2] * rs, acc[0][3] * rs};
    float up[4] = {0.f, 0.f, 0.f, 0.f};            // SwiGLU: the "up" half
    if constexpr (DUAL) { up[0] = acc[1][0] * rs; up[1] = acc[1][1] * rs; up[2] = acc[1][2] * rs; up[3] = acc[1][3] * rs; }
    if constexpr (NOP == 2) {                      // + W.shift
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[r] += acc[NM][r]; if constexpr (DUAL) up[r] += acc[NM + 1][r]; }
    }
    const float pb[4] = {pre_b.x, pre_b.y, pre_b.z, pre_b.w};
    const float py[4] = {(pre_y.x + pre_y0.x) + pre_y1.x, (pre_y.y + pre_y0.y) + pre_y1.y, (pre_y.z + pre_y0.z) + pre_y1.z, (pre_y.w + pre_y0.w) + pre_y1.w};
    const float pg[4] = {pre_g.x, pre_g.y, pre_g.z, pre_g.w};
    if constexpr (EPI == VV_EPI_BIAS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += pb[r];
    } else if constexpr (EPI == VV_EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = gelu_erf_f(o[r] + pb[r]);
    } else if constexpr (EPI == VV_EPI_SWIGLU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = silu_acc(o[r]) * up[r];
    } else if constexpr (EPI == VV_EPI_RESID) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (ksb == 0) ? py[r] + pg[r] * (o[r] + pb[r]) : pg[r] * o[r];
    } else if constexpr (EPI == VV_EPI_GATED_RESID) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (ksb == 0) ? py[r] + pg[r] * o[r] : pg[r] * o[r];
    }
    if constexpr (EPI == VV_EPI_CFG_DPM) {
        const int nc = a.n_cfg;
        const float ca = a.coef[0], cs_ = a.coef[1], csx = a.coef[2], c0 = a.coef[3], c1 = a.coef[4];
        const float cn = a.sde_noise ? a.coef[5] : 0.f;          // sde-dpmsolver++: + cn * eps_i (dpm_solver.py:680-686, 785-793)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float vu = __shfl(o[r], lane + nc);
            const int n = n0 + r;
            if (frow < nc && n < pN) {
                const float v = vu + a.cfg * (o[r] - vu);
                const unsigned zi = (unsigned)(frow * pN + n);
                const float zo = a.z[zi];
                const float x0 = ca * zo - cs_ * v;
                float zn = csx * zo + c0 * x0 + c1 * (x0 - a.x0p[zi]);
                if (a.sde_noise) zn += cn * a.sde_noise[zi];
                if constexpr (CHAIN) {           // write-through: the next phase's workgroups sit on other XCDs
                    float* const zw = a.z_out ? a.z_out : a.z;
                    vv_chain_store1((a.x0p_out ? a.x0p_out : a.x0p) + zi, x0); vv_chain_store1(zw + zi, zn); vv_chain_store1(zw + zi + (unsigned)(nc * pN), zn);
                } else {
                    a.x0p[zi] = x0;
                    a.z[zi] = zn;
                    a.z[zi + (unsigned)(nc * pN)] = zn;
                }
            }
        }
        if constexpr (CHAIN) vv_chain_publish(cs);
        return;
    }
    float* yp = (ksb == 0 ? pY : a.yparts + (unsigned)((ksb - 1) * a.part_stride)) + (yrow_off + (unsigned)n0);
    if constexpr (CHAIN) {
        vv_chain_store4(yp, f32x4{o[0], o[1], o[2], o[3]});
        vv_chain_publish(cs);
    } else {
        *reinterpret_cast<float4*>(yp) = float4{o[0], o[1], o[2], o[3]};
    }
    VV_STAMP(6);
    VV_BSTAMP(1);
}


}  // namespace

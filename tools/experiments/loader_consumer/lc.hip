// lc.hip -- EXPERIMENT (round 4; not part of libvvhip.so): the diffusion head's MLP chain as ONE persistent launch on a
// loader / consumer engine, against the same chain as dependent GEMV launches (the product's form).
//
//   chain      S solver steps x L layers x { A: u = silu(rs * Wg.x) * (rs * Wu.x)   (RMSNorm folded: rs = rsqrt(mean x^2 + eps))
//                                            B: x = x + Wd.u }                      2 activation rows (cond / uncond)
//   engine     one 256-thread workgroup per CU (grid = 256), wave 0 = LOADER, waves 1-3 = CONSUMERS
//     loader   streams this CU's weight rows -- a per-CU contiguous byte stream laid out in consumption order (bench_lc.py) --
//              through an NSLOT x 16 KiB LDS ring with global_load_lds (1 KiB per instruction, nt policy); runs ahead of the
//              dependency by up to the ring's size (that is the point: a launch chain restarts its weight stream at every
//              kernel boundary, the loader never stops)
//     consumer each wave owns feature PAIRS of the op (4 weight rows for A, 2 for B), reads 1 KiB chunks from the ring,
//              v_dot2c_f32_bf16 against the activation vector (bf16, LDS), DPP wave reduction, epilogue, publishes the
//              pair as ONE 8-byte {epoch, bf16 pair} granule per row (write-through sc1 store: the data is the flag)
//     hand-off every CU needs the whole activation vector of the next op: its consumer waves sweep the granule array with
//              relaxed agent-scope loads until every tag equals the epoch (cdna_hip_programming.md Guideline 16, recipe R2),
//              writing the values straight into the LDS activation vector; no fence, no flag, no grid barrier
//   every spin is bounded; a timeout sets a global abort word and the wave leaves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

struct LCParams {
    const unsigned char* wstream;   // [256 CUs][L * CPL chunks][1 KiB]
    u64* xg;                        // granules of the H-vector: index r * (H/2) + pair
    u64* ug;                        // granules of the F-vector
    const float* x_init;            // [2][H]
    float* x_out;                   // [2][H]
    unsigned* abort_word;           // 0 = ok
    u64* stamps;                    // optional [256][8] wall-clock stamps (100 MHz)
    int L, S;
    float eps;
};

namespace {

constexpr unsigned SPIN_LIMIT = 400000;     // x s_sleep(4) ~ 256 cycles each: ~40 ms, then abort

__device__ __forceinline__ float wave_sum(float v) {
    int x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));
    x = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float bf_lo_(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi_(unsigned v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }
#ifdef LC_FMA
__device__ __forceinline__ float dot8(const u32x4 w, const u32x4 a, float c) {
    c += bf_lo_(w.x) * bf_lo_(a.x) + bf_hi_(w.x) * bf_hi_(a.x);
    c += bf_lo_(w.y) * bf_lo_(a.y) + bf_hi_(w.y) * bf_hi_(a.y);
    c += bf_lo_(w.z) * bf_lo_(a.z) + bf_hi_(w.z) * bf_hi_(a.z);
    c += bf_lo_(w.w) * bf_lo_(a.w) + bf_hi_(w.w) * bf_hi_(a.w);
    return c;
}
#else
__device__ __forceinline__ float dot2(unsigned w, unsigned a, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w), __builtin_bit_cast(bf16x2, a), c, false);
}
__device__ __forceinline__ float dot8(const u32x4 w, const u32x4 a, float c) {
    // NB: the elements go through `unsigned` temporaries: __builtin_bit_cast(bf16x2, w.y) applied to a vector ELEMENT directly reads
    // element 0 (clang, ROCm 7.2 -- the ISA showed four v_dot2c on the same registers; same quirk as DESIGN.md section 8 notes for
    // the permlane swap's result)
    const unsigned w0 = w.x, w1 = w.y, w2 = w.z, w3 = w.w, a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w;
    c = dot2(w0, a0, c); c = dot2(w1, a1, c); c = dot2(w2, a2, c); c = dot2(w3, a3, c);
    return c;
}
#endif
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ float bf_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }
__device__ __forceinline__ float silu(float u) { return u / (1.0f + __expf(-u)); }

struct Ctrl {                       // LDS control block
    volatile unsigned landed;       // slots whose bytes are in LDS
    volatile unsigned slot_cnt[8];  // chunks consumed from each ring slot, cumulative
    volatile unsigned gather_cnt;   // consumer waves that finished a gather, cumulative
    volatile unsigned abort_l;
    volatile unsigned gathering;    // consumer waves inside a gather (the loader thins itself meanwhile)
    float ssq[3][2];
};

template <int H, int F, int NSLOT>
__global__ __launch_bounds__(256, 1) void lc_chain_kernel(const LCParams p) {
    constexpr int CA = H / 512, CB = F / 512;        // 1 KiB chunks per weight row
    constexpr int nA = F / 256, nB = H / 256;        // features of A's / B's output owned by one CU
    constexpr int PA = nA / 2, PB = nB / 2;          // feature pairs
    constexpr int CPL = PA * 4 * CA + PB * 2 * CB;   // chunks per layer and CU
    static_assert(H % 512 == 0 && F % 512 == 0 && nA % 2 == 0 && nB % 2 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* ring = lds;
    unsigned* actx = reinterpret_cast<unsigned*>(lds + NSLOT * 16384);            // [2][H/2] bf16 pairs
    unsigned* actu = actx + H;                                                     // [2][F/2]
    float* xown = reinterpret_cast<float*>(actu + F);                              // [2][nB]
    Ctrl* ct = reinterpret_cast<Ctrl*>(xown + 2 * nB + 2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cu = blockIdx.x;
    if (tid == 0) {
        ct->landed = 0; ct->gather_cnt = 0; ct->abort_l = 0; ct->gathering = 0;
        for (int i = 0; i < 8; ++i) ct->slot_cnt[i] = 0;
        for (int i = 0; i < 3; ++i) { ct->ssq[i][0] = 0.f; ct->ssq[i][1] = 0.f; }
    }
    __syncthreads();
    const int n_ops = p.S * p.L * 2;
    // the per-CU stream of ONE solver step is padded to whole ring slots (16 chunks): the loader never wraps inside a slot
    const int step_chunks = p.L * CPL, step_slots = (step_chunks + 15) / 16, step_pad = step_slots * 16 - step_chunks;
    const int total_slots = p.S * step_slots;

    auto fail = [&](unsigned code) {
        ct->abort_l = code;
        __hip_atomic_store(p.abort_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    if (wave == 0) {
        // ------------------------------------------------------------------ LOADER
        const unsigned char* base = p.wstream + (size_t)cu * ((size_t)step_slots * 16384) + lane * 16;
        unsigned pub = 0;                               // slots published as landed (monotonic)
        u64 t_space = 0, t_vm = 0;
        const u64 t_begin = __builtin_amdgcn_s_memrealtime();
        int ks = 0;                                     // slot inside the step
        for (int k = 0; k < total_slots; ++k) {
            const int slot = k % NSLOT;
            const unsigned need = 16u * (unsigned)(k / NSLOT);
            const u64 ts0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned spins = 0; ct->slot_cnt[slot] < need; ++spins) {
                if (spins == 0) {                      // blocked on ring space anyway: everything issued so far may as well be published
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if ((unsigned)k > pub) { pub = (unsigned)k; if (lane == 0) ct->landed = pub; }
                }
                __builtin_amdgcn_s_sleep(2);
                if (ct->abort_l) return;
                if (spins > SPIN_LIMIT) { if (lane == 0) fail(0x100u + (unsigned)slot); return; }
            }
            const u64 ts1 = __builtin_amdgcn_s_memrealtime();
            t_space += ts1 - ts0;
            // 16 straight-line 1 KiB copies (no per-load bookkeeping: the first version's 64-bit modulo per load, then a compare and
            // branch per load, made the LOADER the bottleneck -- 7.5 and 13 GB/s per CU)
            const unsigned char* src = base + (size_t)ks * 16384;
            unsigned char* dst = ring + slot * 16384;
#pragma unroll
            for (int c = 0; c < 16; ++c)
                __builtin_amdgcn_global_load_lds((gvoid_t*)(src + c * 1024), (lvoid_t*)(dst + c * 1024), 16, 0, 2);
            if (++ks == step_slots) {                   // the step's last slot: its padding chunks are nobody's to consume
                ks = 0;
                if (step_pad && lane == 0) atomicAdd((unsigned*)&ct->slot_cnt[slot], (unsigned)step_pad);
            }
            const u64 ts2 = __builtin_amdgcn_s_memrealtime();
            if (ct->gathering) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this CU's consumers are sweeping granules: one fill in flight
            asm volatile("s_waitcnt vmcnt(47)" ::: "memory");           // at most 47 copies outstanding: slot k-3 has landed
            t_vm += __builtin_amdgcn_s_memrealtime() - ts2;
            if (k >= 3 && (unsigned)(k - 2) > pub) { pub = (unsigned)(k - 2); if (lane == 0) ct->landed = pub; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) ct->landed = (unsigned)total_slots;
        if (p.stamps && lane == 0 && (cu == 0 || cu == 131)) {
            u64* q = p.stamps + (cu == 0 ? 0 : 8);
            q[0] = __builtin_amdgcn_s_memrealtime() - t_begin; q[1] = t_space; q[2] = t_vm; q[3] = (u64)total_slots;
        }
        return;
    }

    // ---------------------------------------------------------------------- CONSUMERS
    const int cw = wave - 1;
    // publish this CU's slice of the initial x (epoch 1) and keep it as the fp32 residual
    for (int pr = cw; pr < PB; pr += 3) {
        const int h = cu * nB + 2 * pr;
        if (lane < 2) {
            const float a = p.x_init[lane * H + h], b = p.x_init[lane * H + h + 1];
            xown[lane * nB + 2 * pr] = a; xown[lane * nB + 2 * pr + 1] = b;
            __hip_atomic_store(p.xg + lane * (H / 2) + (h >> 1), ((u64)1 << 32) | pack2(a, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    long chunk_base = 0;                               // first chunk of the current op in this CU's (padded) stream
    for (int op = 0; op < n_ops; ++op) {
        const bool isA = (op & 1) == 0;
        const unsigned epoch_in = (unsigned)op + 1u, epoch_out = (unsigned)op + 2u;
        const bool stamp = p.stamps && cw == 0 && lane == 0 && (cu == 0 || cu == 131) && op < 64;
        u64* stp = p.stamps + 64 + ((cu == 0 ? 0 : 64) + op) * 4;
        if (stamp) stp[0] = __builtin_amdgcn_s_memrealtime();
        // ---- gather the op's input vector: x (H) for A, u (F) for B ----
        {
            const int NG = isA ? H : F;                 // granules = pairs x 2 rows
            u64* g = isA ? p.xg : p.ug;
            unsigned* act = isA ? actx : actu;
            float ss0 = 0.f, ss1 = 0.f;
#ifndef LC_NO_THIN
            if (lane == 0) atomicAdd((unsigned*)&ct->gathering, 1u);
#endif
            // this wave's batches (512 granules each): b = cw, cw + 3, ...  ALL of them are requested in every pass (one latency
            // per pass, not one per batch); a batch whose 512 tags all match is written to LDS and leaves the pending set
            constexpr int NBW = ((F > H ? F : H) / 512 + 2) / 3;          // batches per wave, at most
            const int nb_tot = NG / 512;
            unsigned pending = 0;
            for (int i = 0; i < NBW; ++i) if (cw + 3 * i < nb_tot) pending |= 1u << i;
            for (unsigned spins = 0; pending; ++spins) {
                u64 gv[NBW][8];
#pragma unroll
                for (int i = 0; i < NBW; ++i) {
                    if (pending & (1u << i)) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            gv[i][j] = __hip_atomic_load(g + (cw + 3 * i) * 512 + j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int i = 0; i < NBW; ++i) {
                    if (pending & (1u << i)) {
                        bool ok = true;
#pragma unroll
                        for (int j = 0; j < 8; ++j) ok &= (unsigned)(gv[i][j] >> 32) == epoch_in;
                        if (__all(ok)) {
                            pending &= ~(1u << i);
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int idx = (cw + 3 * i) * 512 + j * 64 + lane;
                                const unsigned v = (unsigned)gv[i][j];
                                act[idx] = v;
                                if (isA) {
                                    const float lo = bf_lo(v), hi = bf_hi(v);
                                    if (idx < NG / 2) ss0 += lo * lo + hi * hi; else ss1 += lo * lo + hi * hi;
                                }
                            }
                        }
                    }
                }
                if (pending) {
                    __builtin_amdgcn_s_sleep(1);
                    if (ct->abort_l) return;
                    if (spins > SPIN_LIMIT) { if (lane == 0) fail(0x200u + (unsigned)op); return; }
                }
            }
            if (isA) {
                ss0 = wave_sum(ss0); ss1 = wave_sum(ss1);
                if (lane == 0) { atomicAdd(&ct->ssq[op % 3][0], ss0); atomicAdd(&ct->ssq[op % 3][1], ss1); }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's LDS writes are ordered before its report
            if (stamp) stp[1] = __builtin_amdgcn_s_memrealtime();
#ifndef LC_NO_THIN
            if (lane == 0) atomicSub((unsigned*)&ct->gathering, 1u);
#endif
            if (lane == 0) atomicAdd((unsigned*)&ct->gather_cnt, 1u);
            const unsigned want = 3u * ((unsigned)op + 1u);
            for (unsigned spins = 0; ct->gather_cnt < want; ++spins) {
                __builtin_amdgcn_s_sleep(1);
                if (ct->abort_l) return;
                if (spins > SPIN_LIMIT) { if (lane == 0) fail(0x300u + (unsigned)op); return; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // the other waves' activation words / sums are read below
            if (stamp) stp[2] = __builtin_amdgcn_s_memrealtime();
            if (cw == 0 && lane == 0) { ct->ssq[(op + 2) % 3][0] = 0.f; ct->ssq[(op + 2) % 3][1] = 0.f; }
        }
        float rs0 = 1.f, rs1 = 1.f;
        if (isA) {
            rs0 = rsqrtf(*(volatile float*)&ct->ssq[op % 3][0] / (float)H + p.eps);
            rs1 = rsqrtf(*(volatile float*)&ct->ssq[op % 3][1] / (float)H + p.eps);
        }
        // ---- compute this wave's feature pairs ----
        const int NP = isA ? PA : PB, RPP = isA ? 4 : 2, CPR = isA ? CA : CB;
        bool dead = false;
        // one weight row: its CPR_ chunks read from the ring in ONE batch (the row's last chunk has landed -> all of it has), then
        // the dot products against both activation rows; the ring slots are released after the reads
        auto row_dot = [&](long c0, auto cpr_c, const unsigned* act, int K, float& o0, float& o1) {
            constexpr int CPR_ = decltype(cpr_c)::value;
            const unsigned need = (unsigned)((c0 + CPR_ - 1) >> 4) + 1u;
            for (unsigned spins = 0; ct->landed < need; ++spins) {
                __builtin_amdgcn_s_sleep(1);
                if (ct->abort_l) { dead = true; return; }
                if (spins > SPIN_LIMIT) { if (lane == 0) fail(0x400u + (unsigned)op); dead = true; return; }
            }
            asm volatile("" ::: "memory");       // compiler barrier only: the LDS executes a wave's operations in order, and the loader's
                                                 // `landed` store follows its own vmcnt wait (a workgroup acquire fence here also waited for this
                                                 // wave's write-through granule stores: ~1 us per feature pair)
            u32x4 w[CPR_];
            const unsigned cb = (unsigned)c0;
#pragma unroll
            for (int j = 0; j < CPR_; ++j) {
                const unsigned ci = cb + (unsigned)j;
                w[j] = *reinterpret_cast<const u32x4*>(ring + ((ci >> 4) % NSLOT) * 16384 + (ci & 15u) * 1024 + lane * 16);
            }
            float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};       // four independent chains per activation row
#pragma unroll
            for (int j = 0; j < CPR_; ++j) {
                const u32x4 x0 = *reinterpret_cast<const u32x4*>(act + j * 256 + lane * 4);
                const u32x4 x1 = *reinterpret_cast<const u32x4*>(act + K / 2 + j * 256 + lane * 4);
                const unsigned w0 = w[j].x, w1 = w[j].y, w2 = w[j].z, w3 = w[j].w;
                const unsigned p0 = x0.x, p1 = x0.y, p2 = x0.z, p3 = x0.w, q0 = x1.x, q1 = x1.y, q2 = x1.z, q3 = x1.w;
                acc[0][0] = dot2(w0, p0, acc[0][0]); acc[1][0] = dot2(w0, q0, acc[1][0]);
                acc[0][1] = dot2(w1, p1, acc[0][1]); acc[1][1] = dot2(w1, q1, acc[1][1]);
                acc[0][2] = dot2(w2, p2, acc[0][2]); acc[1][2] = dot2(w2, q2, acc[1][2]);
                acc[0][3] = dot2(w3, p3, acc[0][3]); acc[1][3] = dot2(w3, q3, acc[1][3]);
            }
            const float a0 = acc[0][0] + acc[0][1], b0 = acc[0][2] + acc[0][3], a1 = acc[1][0] + acc[1][1], b1 = acc[1][2] + acc[1][3];
            // release the ring slots (after the reads; a wave's LDS operations execute in order): one add per slot touched
            asm volatile("" ::: "memory");
            if (lane == 0) {
                const unsigned cl = cb + CPR_ - 1;
                for (unsigned sl = cb >> 4; sl <= (cl >> 4); ++sl) {
                    const unsigned lo = sl * 16 > cb ? sl * 16 : cb, hi = sl * 16 + 15 < cl ? sl * 16 + 15 : cl;
                    atomicAdd((unsigned*)&ct->slot_cnt[sl % NSLOT], hi - lo + 1);
                }
            }
            o0 = wave_sum(a0 + b0);
            o1 = wave_sum(a1 + b1);
        };
        using CA_t = std::integral_constant<int, CA>;
        using CB_t = std::integral_constant<int, CB>;
        for (int pr = cw; pr < NP; pr += 3) {
            const long c0 = chunk_base + (long)pr * RPP * CPR;
            if (isA) {
                float g0a, g0b, u0a, u0b, g1a, g1b, u1a, u1b;
                row_dot(c0, CA_t{}, actx, H, g0a, g0b); if (dead) return;
                row_dot(c0 + CA, CA_t{}, actx, H, u0a, u0b); if (dead) return;
                row_dot(c0 + 2 * CA, CA_t{}, actx, H, g1a, g1b); if (dead) return;
                row_dot(c0 + 3 * CA, CA_t{}, actx, H, u1a, u1b); if (dead) return;
                const int f = cu * nA + 2 * pr;
                if (lane < 2) {
                    const float rs = lane == 0 ? rs0 : rs1;
                    const float g0 = (lane == 0 ? g0a : g0b) * rs, u0 = (lane == 0 ? u0a : u0b) * rs;
                    const float g1 = (lane == 0 ? g1a : g1b) * rs, u1 = (lane == 0 ? u1a : u1b) * rs;
                    const unsigned val = pack2(silu(g0) * u0, silu(g1) * u1);
                    __hip_atomic_store(p.ug + lane * (F / 2) + (f >> 1), ((u64)epoch_out << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                float d0a, d0b, d1a, d1b;
                row_dot(c0, CB_t{}, actu, F, d0a, d0b); if (dead) return;
                row_dot(c0 + CB, CB_t{}, actu, F, d1a, d1b); if (dead) return;
                const int h = cu * nB + 2 * pr;
                if (lane < 2) {
                    const float n0 = xown[lane * nB + 2 * pr] + (lane == 0 ? d0a : d0b);
                    const float n1 = xown[lane * nB + 2 * pr + 1] + (lane == 0 ? d1a : d1b);
                    xown[lane * nB + 2 * pr] = n0; xown[lane * nB + 2 * pr + 1] = n1;
                    p.x_out[lane * H + h] = n0; p.x_out[lane * H + h + 1] = n1;
                    __hip_atomic_store(p.xg + lane * (H / 2) + (h >> 1), ((u64)epoch_out << 32) | pack2(n0, n1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (stamp) stp[3] = __builtin_amdgcn_s_memrealtime();
        chunk_base += (long)NP * RPP * CPR;
        if ((op + 1) % (2 * p.L) == 0) chunk_base += step_pad;          // end of a solver step: skip the slot padding
    }
}

}  // namespace

template <int H, int F, int NSLOT>
static int go(const LCParams& p, hipStream_t s) {
    constexpr size_t smem = (size_t)NSLOT * 16384 + (size_t)(H + F) * 4 + (size_t)(2 * (H / 256) + 2) * 4 + sizeof(Ctrl) + 64;
    static_assert(smem <= 160 * 1024, "LDS");
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lc_chain_kernel<H, F, NSLOT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -3;
        attr = true;
    }
    // state re-initialised every call (Guideline 16): granule tags, abort word
    if (hipMemsetAsync(p.xg, 0, (size_t)H * 8, s) != hipSuccess) return -4;
    if (hipMemsetAsync(p.ug, 0, (size_t)F * 8, s) != hipSuccess) return -4;
    if (hipMemsetAsync(p.abort_word, 0, 4, s) != hipSuccess) return -4;
    hipLaunchKernelGGL((lc_chain_kernel<H, F, NSLOT>), dim3(256), dim3(256), smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int lc_run(void* stream, int H, int F, const void* wstream, void* xg, void* ug, const float* x_init, float* x_out,
                      unsigned* abort_word, void* stamps, int L, int S, float eps) {
    LCParams p;
    p.wstream = (const unsigned char*)wstream; p.xg = (u64*)xg; p.ug = (u64*)ug; p.x_init = x_init; p.x_out = x_out;
    p.abort_word = abort_word; p.stamps = (u64*)stamps; p.L = L; p.S = S; p.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    if (H == 1536 && F == 4608) return go<1536, 4608, 8>(p, s);
    if (H == 3584 && F == 10752) return go<3584, 10752, 5>(p, s);
    if (H == 1024 && F == 3072) return go<1024, 3072, 8>(p, s);          // small test shape
    return -1;
}

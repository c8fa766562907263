#!/usr/bin/env python3
"""Generates attn_stamped.hip from the product's attn.hip: the same source with wall-clock stamps (s_memrealtime, 100 MHz, one
clock for the whole chip) inserted into vv_attn_fused_kernel and vv_attn_merge2_kernel, written to a global `g_stamps` array by
wave 0 / lane 0 of every workgroup of row 0.  The product file is not touched (its hash is the library's build id)."""
import os, re, sys
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "../../../vibevoice_amd/csrc/attn.hip")).read()

def ins_after(s, anchor, text, count=1):
    i = s.index(anchor)
    j = i + len(anchor)
    return s[:j] + text + s[j:]

def ins_before(s, anchor, text):
    i = s.index(anchor)
    return s[:i] + text + s[i:]

s = src
# global stamp buffer + helper
s = s.replace('#include "vv_common.h"', '#include "vv_common.h"\n__device__ unsigned long long* g_stamps = nullptr;\n'
              '#define STAMP(slot_, i_) do { if (g_stamps && blockIdx.z == 0 && threadIdx.x == 0) '
              'g_stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 16 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)\n', 1)
k0 = s.index("__global__ __launch_bounds__(WAVES * 64) void vv_attn_fused_kernel(")
k1 = s.index("// grid (R, Hq), block D threads")
body = s[k0:k1]
body = ins_after(body, "    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n", "    STAMP(0, 0);\n")
body = ins_after(body, "    if (split >= used) return;\n", "    STAMP(0, 1);                              // row table read (pos known)\n")
body = ins_after(body, "    if (p_first < end) kv_load(p_first);\n", "    STAMP(0, 2);                              // first K/V block requested\n")
body = ins_after(body, "    __syncthreads();                         // knew / vnew visible to the wave that meets the new token\n",
                 "    STAMP(0, 3);                              // q fragments ready\n    int it_ = 0;\n")
body = ins_after(body, "                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vt[dt]), pb[p], o[dt], 0, 0, 0);\n        }\n",
                 "        if (it_ == 0) { asm volatile(\"\" :: \"v\"(o[0][0])); STAMP(0, 4); }   // first block consumed\n        ++it_;\n")
body = ins_before(body, "    lsum += __shfl_xor(lsum, 16);\n", "    asm volatile(\"\" :: \"v\"(o[0][0])); STAMP(0, 5);      // stream done (this wave)\n")
body = ins_after(body, "    for (int dt = 0; dt < DT; ++dt) so[wave][dt][lane] = o[dt];\n    __syncthreads();\n", "    STAMP(0, 6);                              // all waves done\n")
# end of kernel: before the final closing of `if (wave == 0) {...}` -> append at the very end of the kernel body
end = body.rindex("}\n")
body = body[:end] + "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); STAMP(0, 7);   // partials stored\n" + body[end:]
s = s[:k0] + body + s[k1:]
# merge2: entry and exit stamps into slots 8, 9 of workgroup (h, r=0)
m0 = s.index("__global__ __launch_bounds__(512) void vv_attn_merge2_kernel(")
mb = s.index("{", s.index(")", m0)) + 1
s = s[:mb] + "\n    if (g_stamps && blockIdx.x == 0 && threadIdx.x == 0) g_stamps[(size_t)4096 * 16 + blockIdx.y * 2] = __builtin_amdgcn_s_memrealtime();\n" + s[mb:]
# exit stamp: at the end of merge2 kernel
m_end = s.index("\n}\n", mb)
s = s[:m_end] + "\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    if (g_stamps && blockIdx.x == 0 && threadIdx.x == 0) g_stamps[(size_t)4096 * 16 + blockIdx.y * 2 + 1] = __builtin_amdgcn_s_memrealtime();" + s[m_end:]
s += '\nextern "C" void vv_exp_set_stamps(unsigned long long* p) { hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)); }\n'
open(os.path.join(here, "attn_stamped.hip"), "w").write(s)
print("wrote attn_stamped.hip", len(s))

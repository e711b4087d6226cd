// parakeet.cpp_amd/csrc/kernels/decode_persist.hip -- the whole TDT / RNNT greedy loop of a batch in ONE kernel launch
// (reference loops: tdt_greedy_decode src/tdt.cpp:62-106, rnnt_greedy_decode src/rnnt.cpp:76-109).
//
// Why: the per-phase version (Model::run_tdt: LSTM cell GEMV -> joint activation GEMV -> heads GEMV -> decide, 4+ launches per
// symbol step, ~512 per 64-clip batch) is bound by launch latency (~12 us per launch around a 2.7 us dependent MFMA chain), and --
// measured in round 2, tools/experiments/launch_interference.py -- every small launch on the decode stream costs the encoder of the
// NEXT batch, which runs concurrently on the other stream, 2.5-6 us (kernel boundaries carry cache release / acquire work): 1.3 ms of
// a 20 ms step.  Here the batch is decoded by one persistent grid:
//  * G = Hp/4 workgroups of 256 threads (160 for Hp = 640: one per LSTM gate-column tile), all co-resident (one per CU is enough).
//  * The phases of a step run back to back inside the kernel, separated by a grid barrier (XCD-hierarchical since round 4: per-XCC arrival
//    counters under a top counter, relaxed atomics, s_sleep while polling).  NO agent-scope fence is used (it would write back and invalidate the
//    XCD's whole L2 under the concurrently running encoder GEMMs): the few words the workgroups exchange (h', c', z, logits, token,
//    frame index, flags) are written and read with system-scope accesses (sc0 sc1), and a workgroup arrives at the barrier only after
//    s_waitcnt vmcnt(0) has confirmed its stores.
//  * The arithmetic is the per-phase kernels' own device code (decode_dev.hpp: skinny_tile, tdt_decide_one) -- same single-chain
//    16x16x4 MFMA products in natural k order, same decide: token ids, frames, confidences are bit-identical to the per-phase path.
//  * A barrier that does not complete within ~2 s (it cannot, short of a fault) raises an abort flag that every workgroup honours:
//    unfinished utterances are reported with length -1 (PK_ERR_DECODE_CAP) instead of hanging the GPU.
#include "decode_dev.hpp"

namespace pk {

template <int NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void tdt_persistent_kernel(TdtPersist p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // decide scratch; the GEMV phases use its first 4*16*17 floats
    __shared__ int s_ctl;
    float(*tile)[16][17] = reinterpret_cast<float(*)[16][17]>(sm);
    const int wg = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    const TdtState &st = p.st;
    const int B = st.B;
    const int n_mg = (B + 63) / 64;
    unsigned phase = 0;
    bool aborted = false;

    // Grid barrier, XCD-hierarchical (MI355X_MICROARCH.md "barrier-xcd"; measured with tools/ubench/grid_barrier.cpp on an idle chip at 160
    // workgroups: 2.7 us against 3.2 us for one flat counter, 5.3 against 6.4 us with 3-9 us of skewed phase work): every workgroup arrives on
    // the counter of ITS XCC (8 independent words instead of one: the arrivals no longer serialise on a single address), the XCC's last
    // arriver goes up to a top counter that sees 8 arrivals, waits there for all XCCs and then bumps its XCC's generation word, which the other
    // workgroups of the XCC poll.  Where a workgroup runs is READ (HW_REG_XCC_ID), never assumed: a census of workgroups per XCC is taken at
    // kernel start behind one flat barrier.  All words are agent / system-scope atomics at the memory side (workgroup-scope atomics are not
    // coherent between the CUs of an XCC: measured, the barrier then never completes); no cache-wide fence is involved -- the exchanged
    // vectors keep travelling through system-scope accesses (see the header).
    unsigned *const bw = p.bar;                                      // [0] flat (census barrier), [32] top, [64 + 32 x] census / arrivals / generation of XCC x
    __shared__ unsigned s_x, s_nx, s_nxcc;
    auto spin = [&](auto cond) {                                     // bounded poll: true = gave up (abort raised here or elsewhere)
        const long long t0 = wall_clock64();
        while (cond()) {
            __builtin_amdgcn_s_sleep(1);
            if (__hip_atomic_load(p.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return true;
            if (wall_clock64() - t0 > p.timeout_ticks) {
                __hip_atomic_store(p.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return true;
            }
        }
        return false;
    };
    if (tid == 0) {
        const unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;     // HW_REG_XCC_ID
        s_x = x;
        __hip_atomic_fetch_add(bw + 64 + 96 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(bw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        int ab = spin([&] { return __hip_atomic_load(bw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)G; }) ? 1 : 0;
        unsigned nx = 0;
        for (int i = 0; i < 8; ++i) nx += __hip_atomic_load(bw + 64 + 96 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? 1u : 0u;
        s_nx = __hip_atomic_load(bw + 64 + 96 * x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_nxcc = nx;
        s_ctl = ab;
    }
    __syncthreads();
    if (s_ctl) aborted = true;
    unsigned *const xarr = bw + 64 + 96 * s_x + 32, *const xgen = xarr + 32;
    const unsigned n_x = s_nx, n_xcc = s_nxcc;

    auto grid_barrier = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): this thread's system-scope stores are performed
        __syncthreads();
        ++phase;
        if (tid == 0) {
            int ab = 0;
            const unsigned old = __hip_atomic_fetch_add(xarr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == phase * n_x) {                            // this XCC's last arriver
                __hip_atomic_fetch_add(bw + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ab = spin([&] { return __hip_atomic_load(bw + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < phase * n_xcc; }) ? 1 : 0;
                __hip_atomic_fetch_add(xgen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (also when aborting: the pollers see the flag anyway)
            } else {
                ab = spin([&] { return __hip_atomic_load(xgen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase; }) ? 1 : 0;
            }
            s_ctl = ab;
        }
        __syncthreads();
        if (s_ctl) aborted = true;
    };

    for (int step = 0; step < st.max_steps && !aborted; ++step) {
        // 1. prediction net: LSTM layers (upper layers: input projection of the layer below's h' first).  The layer index is a
        //    compile-time constant in every copy of the body: the argument records stay in scalar registers.
#pragma unroll
        for (int l = 0; l < 2; ++l) {                     // the launcher admits at most two LSTM layers
            if (l >= p.L || aborted) break;
            if (l > 0 && !p.cell[l].W2) {                 // (input projection not fused into the cell: its own phase)
                const int n_t = (p.ih[l].N + 15) / 16;
                for (int nt = wg; nt < n_t; nt += G)
                    for (int mg = 0; mg < n_mg; ++mg) skinny_tile<SK_BIAS, NCH, true>(p.ih[l], nt, mg, tile);
                grid_barrier();
                if (aborted) break;
            }
            const int n_t = p.cell[l].Hp / 4;
            for (int nt = wg; nt < n_t; nt += G)
                for (int mg = 0; mg < n_mg; ++mg) skinny_tile<SK_CELL, NCH, true>(p.cell[l], nt, mg, tile);
            grid_barrier();
        }
        if (aborted) break;
        // 2. joint: z = relu(enc_proj[t_b] + pred_proj(h'))
        {
            const int n_t = (p.act.N + 15) / 16;
            for (int nt = wg; nt < n_t; nt += G)
                for (int mg = 0; mg < n_mg; ++mg) skinny_tile<SK_ACT, NCH, true>(p.act, nt, mg, tile);
            grid_barrier();
            if (aborted) break;
        }
        // 3. label (+ duration) heads
        {
            const int n_t = (p.heads.N + 15) / 16;
            for (int nt = wg; nt < n_t; nt += G)
                for (int mg = 0; mg < n_mg; ++mg) skinny_tile<SK_BIAS, NCH, true>(p.heads, nt, mg, tile);
            grid_barrier();
            if (aborted) break;
        }
        // 4. greedy decision per utterance: emit / blank, frame advance, LSTM state commit or revert
        for (int b = wg; b < B; b += G) {
            tdt_decide_one<false, true>(st, b, sm);
            __syncthreads();
        }
        grid_barrier();
        if (aborted) break;
        if (tid == 0) s_ctl = __hip_atomic_load(st.done_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= B ? 1 : 0;
        __syncthreads();
        const int fin = s_ctl;
        __syncthreads();
        if (fin) break;
    }
    if (aborted) {                                                   // never expected: report instead of hanging
        for (int b = wg; b < B; b += G)
            if (tid == 0 && !__hip_atomic_load(st.done + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) st.lens[b] = -1;
    }
}

size_t tdt_persistent_lds_bytes(const TdtState &st) {
    const size_t decide = (size_t)(2 * (st.V + st.D) + 16) * sizeof(float), tile = 4 * 16 * 17 * sizeof(float);
    return decide > tile ? decide : tile;
}

void launch_tdt_persistent(const TdtPersist &p, hipStream_t s) {
    const int G = p.cell[0].Hp / 4;
    const size_t lds = tdt_persistent_lds_bytes(p.st);
    const bool k640 = p.cell[0].K == 640 && p.act.K == 640 && p.heads.K == 640;
    if (k640) hipLaunchKernelGGL(tdt_persistent_kernel<10>, dim3(G), dim3(256), lds, s, p);
    else hipLaunchKernelGGL(tdt_persistent_kernel<0>, dim3(G), dim3(256), lds, s, p);
}

}  // namespace pk

// The dense chain of the interpolation head in train(), forward, as ONE kernel (included by pps_rows_train_impl.h inside namespace PPS_NS):
//
//     h1 = table[ids] + Wx (query - pts[ids])     y2 = fc2(relu(h1))     y3 = fc3(relu(y2))     qy = fc_query(relu(y3))
//
// replaces (reference, under autograd and 16-bit autocast): source/poco_model.py:400-409 on [B * Q * 64, 256] rows -- as separate ops
// (pps_head_input_fwd + three pps_rows_layer_fwd launches) every layer reads its input from HBM and writes its output: 7.25 passes over
// [1.28 M, 256] 16-bit tensors per step.  Here a wave carries its 32 rows (two 16-row tiles) through the three layers in registers -- the C/D layout
// of v_mfma_f32_16x16x32 with the channel -> MFMA-row assignment of rows_layer_kernel (a lane holds 16 consecutive channels of its row) IS the B
// operand layout of the next layer once the contraction index is ordered accordingly (k-step (G, h), slot (kg, j) <-> channel 64 G + 16 kg + 8 h + j;
// baked into the packed weights) -- and the raw outputs h1, y2, y3, qy that the backward pass needs are written once each: 3.25 passes, no reads
// but the gather of the per-point table (51 MB, cache-resident).
// The three weight matrices (288 KB as 16-bit A fragments) do not fit the LDS: they stream L2 -> LDS in 9 chunks of 32 KB (64 output channels x
// 256 input channels; four buffers, requested three chunks ahead, one barrier per chunk) shared by the 8 waves of the workgroup, continuously
// across the row units.
// Arithmetic and rounding points are those of the separate kernels (fp32 accumulation, every stored tensor rounded to the storage type, ReLU on the
// rounded value), so the backward pass -- unchanged -- sees the tensors it would have seen.

constexpr int HC_C = 256;               // channels of the head (latent size)
constexpr int HC_HEADS = 64;            // attention heads = outputs of fc_query
constexpr int HC_CHUNKS = 9;            // fc2: 4 groups of 64 output channels, fc3: 4, fc_query: 1
constexpr int HC_CHUNK_FRAGS = 8 * 4 * 64;          // [k-step][block][lane] fragments of 16 bytes = 32 KB
constexpr int HC_UNIT = 256;            // rows of a workgroup pass: 8 waves x 2 tiles x 16 rows
constexpr int HC_REQ_OPS = HC_CHUNK_FRAGS * 16 / (512 * 16);   // global_load_lds_dwordx4 per wave and chunk (512 threads x 16 bytes per instruction)
static_assert(HC_REQ_OPS == 4, "chunk_copy_async<4, 512> below");
constexpr int HC_NBUF = 4;              // LDS weight buffers: a chunk is requested 3 chunks before it is used (a chunk's products take ~0.9 us of a
                                        // SIMD's matrix pipe, an L2 -> LDS copy ~2 us: with two buffers every chunk waited for the next one)

template <int V> struct HcInt { static constexpr int value = V; };

struct HeadChainArgs {
    const uint16_t* table;      // [n, 256]
    const int64_t* ids;         // [rows]
    const float* pts;           // [n, 3]
    const float* query;         // [rows / k, 3]
    const float* wx;            // [256, 3]
    const float* b2; const float* b3; const float* bq;
    const bf16x8* wimg;         // [9][8][4][64] packed by head_chain_pack_kernel
    uint16_t* h1; uint16_t* y2; uint16_t* y3; uint16_t* qy;
    int64_t rows;
    int k;
};

// fp32 master weights -> A fragments.  Fragment (chunk, s, o, lane = (m, kg)): output channel of MFMA row m of block o of the chunk's group, input
// channels 64 (s >> 1) + 16 kg + 8 (s & 1) + [0, 8)
__global__ __launch_bounds__(256) void head_chain_pack_kernel(const float* __restrict__ w2, const float* __restrict__ w3, const float* __restrict__ wq,
                                                             bf16x8* __restrict__ img) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HC_CHUNKS * HC_CHUNK_FRAGS) return;
    const int lane = i & 63, o = (i >> 6) & 3, s = (i >> 8) & 7, chunk = i >> 11;
    const int m = lane & 15, kg = lane >> 4;
    const float* w = chunk < 4 ? w2 : (chunk < 8 ? w3 : wq);
    const int gout = chunk < 8 ? (chunk & 3) : 0;
    const int co = 64 * gout + 16 * (m >> 2) + 4 * o + (m & 3);
    const float* src = w + (int64_t)co * HC_C + 64 * (s >> 1) + 16 * kg + 8 * (s & 1);
    const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
    const u32x4 p = {pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3])};
    img[i] = as_frag(p);
}

// floats of padding in front of channel c's row of the Wx table: the four lane groups of a wave read rows 16 channels apart -- 256 B, the same
// banks, without it
#define HC_WXPAD(c) (((c) >> 4) * 4)
__global__ __launch_bounds__(512, 1) void head_chain_fwd_kernel(const HeadChainArgs a) {
    bf16x8* wbuf = (bf16x8*)smem;                                   // [HC_NBUF][HC_CHUNK_FRAGS]
    float* bs = (float*)(wbuf + HC_NBUF * HC_CHUNK_FRAGS);          // b2 [256], b3 [256], bq [64]
    float* wxs = bs + 2 * HC_C + HC_HEADS;                          // [256][4]: (wx, wy, wz, 0) of a channel in one 16-byte read
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;

    for (int i = threadIdx.x; i < HC_C; i += 512) { bs[i] = a.b2 ? a.b2[i] : 0.f; bs[HC_C + i] = a.b3 ? a.b3[i] : 0.f; }
    for (int i = threadIdx.x; i < HC_HEADS; i += 512) bs[2 * HC_C + i] = a.bq ? a.bq[i] : 0.f;
    for (int i = threadIdx.x; i < HC_C * 4; i += 512) wxs[i + HC_WXPAD(i >> 2)] = (i & 3) < 3 ? a.wx[(i >> 2) * 3 + (i & 3)] : 0.f;
    // chunks 0, 1, 2 -> buffers 0, 1, 2.  Chunks travel L2 -> LDS without passing through registers (global_load_lds_dwordx4, pps_common.h)
#pragma unroll
    for (int c = 0; c < HC_NBUF - 1; ++c)
        pps::chunk_copy_async<4, 512>((const ::f32x4*)(a.wimg + c * HC_CHUNK_FRAGS), (::f32x4*)(wbuf + c * HC_CHUNK_FRAGS));
    pps::stream_wait();
    __syncthreads();
    int cur = 0;                                                     // buffer of the chunk about to be used

    // row, neighbour id and offset to the neighbour of this lane's two rows of a unit -- fetched ONE UNIT AHEAD in two stages (the id right after
    // this unit's gather, the point it names after the first layer): two dependent memory latencies that would otherwise stand in front of every unit
    // with all eight waves waiting
    auto fetch_ids = [&](int64_t u, int64_t (&rw)[2], int64_t (&id)[2]) {
        const int64_t r0 = u * HC_UNIT + wave * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            rw[t] = r0 + 16 * t + n;
            id[t] = a.ids[rw[t] < a.rows ? rw[t] : a.rows - 1];
        }
    };
    // (query and point stay two registers sets until the unit starts: subtracting here would make the wave wait for them -- and, VMEM operations
    // retiring in order, for the sixteen h1 stores in front of them)
    auto fetch_rel = [&](const int64_t (&rw)[2], const int64_t (&id)[2], const uint16_t* (&tr)[2], float (&qv)[2][3], float (&pv)[2][3]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t q = (rw[t] < a.rows ? rw[t] : a.rows - 1) / a.k;
#pragma unroll
            for (int d = 0; d < 3; ++d) { qv[t][d] = a.query[q * 3 + d]; pv[t][d] = a.pts[id[t] * 3 + d]; }
            tr[t] = a.table + id[t] * HC_C;
        }
    };
    int64_t nid[2];
    int64_t rowu[2], nrowu[2];
    const uint16_t* trow[2];
    const uint16_t* ntrow[2];
    float rel[2][3], nq[2][3], np[2][3];
    const int64_t nunits = (a.rows + HC_UNIT - 1) / HC_UNIT;
    if ((int64_t)blockIdx.x < nunits) { fetch_ids(blockIdx.x, nrowu, nid); fetch_rel(nrowu, nid, ntrow, nq, np); }
    for (int64_t u = blockIdx.x; u < nunits; u += gridDim.x) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            rowu[t] = nrowu[t]; trow[t] = ntrow[t];
#pragma unroll
            for (int d = 0; d < 3; ++d) rel[t][d] = nq[t][d] - np[t][d];
        }
        // ---- h1 = table row + Wx rel, stored; relu(h1) = B fragments of fc2.  ALL sixteen pieces of the two table rows are requested before the
        //      first is used (one memory latency per unit, not eight: the compiler cannot hoist a load over the h1 stores of the previous piece, and
        //      every wait it places also waits for the weight chunk requested last), together with the ids of the next unit
        bf16x8 fa[2][8], fb[2][8];
        u32x4 raw[2][8];
        // The wait is for the last epilogue's stores (the chunks requested during the last three chunks of the previous unit are 1-3 chunk times
        // old): the table rows requested next are then the only loads the compiler's own waits in this phase have to count.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) raw[t][s] = *(const u32x4*)(trow[t] + 64 * (s >> 1) + 16 * g + 8 * (s & 1));
        const bool more = u + gridDim.x < nunits;
        if (more) fetch_ids(u + gridDim.x, nrowu, nid);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int c0 = 64 * (s >> 1) + 16 * g + 8 * (s & 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4 q4 = raw[t][s];
                float e[8] = {lo16(q4.x), hi16(q4.x), lo16(q4.y), hi16(q4.y), lo16(q4.z), hi16(q4.z), lo16(q4.w), hi16(q4.w)};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 w = *(const f32x4*)(wxs + (c0 + j) * 4 + HC_WXPAD(c0 + j));
                    e[j] += w[0] * rel[t][0] + w[1] * rel[t][1] + w[2] * rel[t][2];
                    // Opaque on purpose: it keeps the SLP vectoriser from pairing channels j, j + 1.  The paired form reads four Wx rows ahead
                    // (ds_read_b128 x 4, partial lgkmcnt waits) and builds its operand pairs IN the rows' destination registers (v_pk_mov_b32 /
                    // v_mov_b32 into them while later rows are still in flight) -- and, on gfx950, lanes 48-63 of the second wave of a SIMD then
                    // sometimes saw a stale value in one channel: ~1 row unit in 10^4, different from run to run, 50 x more often after an
                    // unrelated change of the epilogue's code.  Measured (profiles/NOTES_r5.md section 3): 165 of 300 launches at the fit batch's size
                    // differed from the first with the paired code, 0 of 18 000 (9 x 10^7 row units) with this line -- also with that epilogue --, and
                    // h1 is bit-identical to pps_head_input_fwd.
                    asm volatile("" : "+v"(e[j]));
                }
                const u32x4 p = {pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])};
#ifndef PPS_HC_NOSTORE
                *(u32x4*)(a.h1 + rowu[t] * HC_C + c0) = p;
#endif
                fa[t][s] = as_frag(relu_bf16x8(p));
            }
        }
        if (more) fetch_rel(nrowu, nid, ntrow, nq, np);             // (the ids have arrived with the table rows)
        // ---- the nine weight chunks.  One chunk: 64 output channels (4 blocks) x 8 k-steps against both tiles; then its epilogue (bias, store,
        //      the fragments of the NEXT layer for k-steps 2 gout, 2 gout + 1)
        auto chunk = [&](const bf16x8 (&in)[2][8], bf16x8 (&out)[2][8], auto layer_c, auto gout_c, const int next_chunk) {
            constexpr int layer = decltype(layer_c)::value, gout = decltype(gout_c)::value;
            // the chunk three ahead -> the buffer the PREVIOUS chunk used (its last readers passed the barrier that ended it)
#ifndef PPS_HC_NODMA
            pps::chunk_copy_async<4, 512>((const ::f32x4*)(a.wimg + (int64_t)next_chunk * HC_CHUNK_FRAGS),
                                          (::f32x4*)(wbuf + ((cur + HC_NBUF - 1) % HC_NBUF) * HC_CHUNK_FRAGS));
#endif
            const bf16x8* wb = wbuf + cur * HC_CHUNK_FRAGS;
            f32x4 acc[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[t][o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const bf16x8 af = wb[(s * 4 + o) * 64 + lane];
#pragma unroll
#ifndef PPS_HC_NOMFMA
                    for (int t = 0; t < 2; ++t) acc[t][o] = PPS_MFMA16(af, in[t][s], acc[t][o], 0, 0, 0);
#else
                    for (int t = 0; t < 2; ++t) if (s == 0) acc[t][o][0] += (float)af[0] + (float)in[t][o][0];
#endif
                }
            // epilogue: lane (row n, g) holds channels 64 gout + 16 g + 4 o + r
            const float* bias = bs + layer * HC_C + 64 * gout + 16 * g;
            uint16_t* dst = layer == 0 ? a.y2 : (layer == 1 ? a.y3 : a.qy);
            const int ld = layer == 2 ? HC_HEADS : HC_C;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                unsigned p[8];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    f32x4 v = acc[t][o] + *(const f32x4*)(bias + 4 * o);
#ifdef PPS_HC_EB        // (the perturbation of the note in the gather phase: scalar epilogue)
                    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#endif
                    p[2 * o] = pack2(v[0], v[1]);
                    p[2 * o + 1] = pack2(v[2], v[3]);
                }
                const u32x4 lo = {p[0], p[1], p[2], p[3]}, hi = {p[4], p[5], p[6], p[7]};
                uint16_t* d = dst + rowu[t] * ld + 64 * gout + 16 * g;
#ifndef PPS_HC_NOSTORE
                *(u32x4*)d = lo;
                *(u32x4*)(d + 8) = hi;
#else
                if (lo.x == 0x12345678u && hi.y == 0x9abcdef0u) *(u32x4*)d = lo;      // (keeps the values alive)
#endif
                if (layer < 2) {
                    out[t][2 * gout] = as_frag(relu_bf16x8(lo));
                    out[t][2 * gout + 1] = as_frag(relu_bf16x8(hi));
                }
            }
            // the NEXT chunk must have landed.  VMEM operations retire in order on gfx9; YOUNGER than its request are, at least, the requests of the
            // two chunks after it: HC_REQ_OPS global_load_lds each, issued by inline assembly, so their number does not depend on the compiler.
            // Waiting until at most 2 HC_REQ_OPS operations are outstanding is therefore safe WHATEVER the compiler makes of the epilogues' stores
            // (rounds 4-5 waited for vmcnt(20) = 8 requests + 12 stores of three epilogues, hand counted -- a build that emitted fewer
            // stores than assumed would have let a wave read a weight buffer before it had landed).  The price of the safe count is that a chunk
            // also waits for the stores of the epilogue BEFORE its own (the youngest 8 operations are this chunk's 4 stores and the request issued
            // at its head): measured 0.60x ms either way at the fit batch's size (profiles/NOTES_r6.md); PPS_HC_VMCNT selects another count for
            // experiments (20: the old hand count; 0: every chunk waits for its own stores).
#ifndef PPS_HC_VMCNT
#define PPS_HC_VMCNT (2 * HC_REQ_OPS)
#endif
            static_assert(PPS_HC_VMCNT >= 0 && PPS_HC_VMCNT < 64, "vmcnt is a 6-bit field");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPS_HC_VMCNT) : "memory");
            __syncthreads();
            cur = (cur + 1) % HC_NBUF;
        };
        chunk(fa, fb, HcInt<0>{}, HcInt<0>{}, 3); chunk(fa, fb, HcInt<0>{}, HcInt<1>{}, 4); chunk(fa, fb, HcInt<0>{}, HcInt<2>{}, 5);
        chunk(fa, fb, HcInt<0>{}, HcInt<3>{}, 6);
        chunk(fb, fa, HcInt<1>{}, HcInt<0>{}, 7); chunk(fb, fa, HcInt<1>{}, HcInt<1>{}, 8); chunk(fb, fa, HcInt<1>{}, HcInt<2>{}, 0);
        chunk(fb, fa, HcInt<1>{}, HcInt<3>{}, 1);
        chunk(fa, fb, HcInt<2>{}, HcInt<0>{}, 2);
    }
}

constexpr size_t head_chain_lds() { return (size_t)HC_NBUF * HC_CHUNK_FRAGS * 16 + (size_t)(2 * HC_C + HC_HEADS + 4 * HC_C + 4 * (HC_C / 16)) * 4; }

size_t pps_head_chain_ws_bytes() { return (size_t)HC_CHUNKS * HC_CHUNK_FRAGS * 16; }

int pps_head_chain_fwd(const void* table, const int64_t* ids, const float* pts, const float* query, int64_t q, int k, const float* wx, const float* w2,
                       const float* b2, const float* w3, const float* b3, const float* wq, const float* bq, void* h1, void* y2, void* y3, void* qy,
                       void* ws, void* stream) {
    if (q < 0 || k < 1) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!table || !ids || !pts || !query || !wx || !w2 || !w3 || !wq || !h1 || !y2 || !y3 || !qy || !ws || ((uintptr_t)ws & 15)) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    static bool ok = allow_lds(head_chain_fwd_kernel, head_chain_lds());
    if (!ok) return PPS_ERR_LAUNCH;
    hipLaunchKernelGGL(head_chain_pack_kernel, dim3((HC_CHUNKS * HC_CHUNK_FRAGS + 255) / 256), dim3(256), 0, st, w2, w3, wq, (bf16x8*)ws);
    HeadChainArgs a{};
    a.table = (const uint16_t*)table; a.ids = ids; a.pts = pts; a.query = query; a.wx = wx; a.b2 = b2; a.b3 = b3; a.bq = bq;
    a.wimg = (const bf16x8*)ws;
    a.h1 = (uint16_t*)h1; a.y2 = (uint16_t*)y2; a.y3 = (uint16_t*)y3; a.qy = (uint16_t*)qy;
    a.rows = q * k; a.k = k;
    const int grid = grid_for((a.rows + HC_UNIT - 1) / HC_UNIT);
    hipLaunchKernelGGL(head_chain_fwd_kernel, dim3(grid), dim3(512), head_chain_lds(), st, a);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

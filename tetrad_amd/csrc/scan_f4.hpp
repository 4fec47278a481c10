// scan_f4.hpp -- the cooperative site scan on ONE 12-byte record per taxon and 32-site lane-step
// Part of the single translation unit tetrad_hip.hip (included inside its anonymous namespace, after scan.hpp).
//
// Row f4 asks for a packed genotype format, 2-bit base + 1-bit missing: that is the `planes3` array (u32 x 3 per taxon
// and 32 sites: missing bits, base bit 0, base bit 1 -- 3 bits per site against 8 in the reference's tmparr,
// write_database.py:157-168).  tq_scan_wg_kernel streams it for a wave's own rows c, d AND their nibble codes (28 bytes
// per taxon and lane-step), because pattern bytes built 8 sites per instruction from nibble words are cheaper than
// pattern bits pulled out of plane words.  This kernel is the row as written: a wave's own rows come as their plane
// records ONLY (3 vector loads per wave-step instead of 5), nothing is parked, and the set-bit walk makes the pattern
// of a COUNTED site (5.6 of a lane's 32 in subsample mode) itself -- the (a,b) byte straight from the workgroup's shared
// image in LDS (one ds_read_u8, as the parked byte before), the (c,d) nibble by four v_bfe_u32 on the plane words it holds.
// Per wave-step that is 3 loads, no park stores and two image reads fewer -- and 15 vector instructions per walk trip
// instead of 6.  That trade pays: the scan of rounds 1-3 was bound by its LDS instructions with vector issue at 38 % of
// the CU's 2-wave-instructions-per-cycle peak (it had been read as 76 % of a peak of 1).  MEASURED (DESIGN.md section 4.1,
// profiles/r04_scan/README.md section 9): LDS instructions -28 %, LDS-array cycles -22 %, c3 subsample scan 5.63 -> 5.32 ms,
// c2 -7 %, c4 -8 % -- once the image's pattern partial is TRANSPOSED ([dword j][lane]: the walk's byte reads are then
// conflict-free; with the lane-contiguous panels of tq_scan_wg_kernel they cost 600 M conflict cycles per dispatch and
// the whole gain).  Default of subsample mode (option "scan_f4" = -1); in full mode the walk has 19 trips per step and the
// form loses against the slot kernels (scan_f4 = 1 selects it there too); parity-tested (tests/test_gpu_configs.py, the
// "hqr" engine of tests/test_gpu_parity.py) and a fuzz option.
#pragma once

// set-bit walk on plane words: for every set bit i of c, hist[abbyte(i) | cd(i)] += 1.  abbyte = (a<<6|b<<4) of site i, read
// from the TRANSPOSED image [dword j = i >> 2][lane] (256-byte rows in a 2 KiB-aligned block, img_ptr = this lane's dword of
// row 0: the address is the bit field [i & 3 : 0-1][lane : 2-7][i >> 2 : 8-10], bank = lane mod 32 whatever the site -- the
// lane-contiguous image of tq_scan_wg_kernel cost this walk 600 M conflict cycles per dispatch); cd = p1c p0c p1d p0d at bit i
__device__ __forceinline__ void walk_planes(uint32_t c, uint32_t p0c, uint32_t p1c, uint32_t p0d, uint32_t p1d,
                                            const uint8_t *img_ptr, uint32_t *hist_ptr)
{
#ifndef TQ_NO_ASM
    const uint32_t img = lds_offset(img_ptr), hist_off = __builtin_amdgcn_readfirstlane(lds_offset(hist_ptr));
    uint32_t t, i, j, b, x0, x1, x2, x3, one = 1u, k703 = 0x703u;
    uint64_t save;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        TQ_WALK_PRIO_ON
        "v_add_co_u32_e32 %[t], vcc, -1, %[c]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 1f\n"
        "0:\n\t"
        "v_ffbl_b32_e32 %[i], %[c]\n\t"
        "v_and_b32_e32 %[c], %[c], %[t]\n\t"
        "v_lshl_or_b32 %[j], %[i], 6, %[i]\n\t"              // i = 4 j + k: j now also at bits 8-10 (k at 0-1)
        "v_bfi_b32 %[j], %[k703], %[j], %[img]\n\t"
        "ds_read_u8 %[b], %[j]\n\t"
        "v_bfe_u32 %[x0], %[p0d], %[i], 1\n\t"
        "v_bfe_u32 %[x1], %[p1d], %[i], 1\n\t"
        "v_bfe_u32 %[x2], %[p0c], %[i], 1\n\t"
        "v_bfe_u32 %[x3], %[p1c], %[i], 1\n\t"
        "v_lshl_or_b32 %[x0], %[x1], 1, %[x0]\n\t"
        "v_lshl_or_b32 %[x2], %[x3], 1, %[x2]\n\t"
        "v_lshl_or_b32 %[x0], %[x2], 2, %[x0]\n\t"           // (c << 2 | d)
        "v_lshl_add_u32 %[x0], %[x0], 2, %[hist]\n\t"
        "v_add_co_u32_e32 %[t], vcc, -1, %[c]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshl_add_u32 %[b], %[b], 2, %[x0]\n\t"
        "ds_add_u32 %[b], %[one]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execnz 0b\n"
        "1:\n\t"
        TQ_WALK_PRIO_OFF
        "s_mov_b64 exec, %[save]"
        : [c] "+v"(c), [t] "=&v"(t), [i] "=&v"(i), [j] "=&v"(j), [b] "=&v"(b), [x0] "=&v"(x0), [x1] "=&v"(x1),
          [x2] "=&v"(x2), [x3] "=&v"(x3), [save] "=&s"(save)
        : [p0c] "v"(p0c), [p1c] "v"(p1c), [p0d] "v"(p0d), [p1d] "v"(p1d), [img] "v"(img), [hist] "s"(hist_off),
          [one] "v"(one), [k703] "s"(k703)
        : "vcc", "memory");
#else
    while (c) {
        const int i = __builtin_ctz(c);
        c &= c - 1;
        const uint32_t ab = img_ptr[(i >> 2) * (WAVE * 4) + (i & 3)];
        const uint32_t cd = (((p1c >> i) & 1u) << 3) | (((p0c >> i) & 1u) << 2) | (((p1d >> i) & 1u) << 1) | ((p0d >> i) & 1u);
        __hip_atomic_fetch_add(hist_ptr + (ab | cd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#endif
}

template <bool SUB, int NW>
__global__ void __launch_bounds__(NW *WAVE)
tq_scan_f4_kernel(DevData d, const uint32_t *__restrict__ quartets, const uint32_t *__restrict__ order, int64_t Q,
                  uint32_t *__restrict__ cm, int64_t xcd_chunk)
{
    static_assert(NW >= 2 && NW <= 8, "waves per workgroup");
    // image of one step: the (a,b) pattern partial ((a<<2)+b)<<4 per site byte, TRANSPOSED [dword j][lane] (only the walk's
    // byte reads touch it), the combined plane record {p0a, p1a, Ma|Mb, (p0a^p0b)|(p1a^p1b)} and the run-begin word per lane
    __shared__ __attribute__((aligned(2048))) uint32_t abp_t[2][8 * WAVE];
    __shared__ uint4 r1_img[2][WAVE];
    __shared__ uint32_t b_img[2][WAVE];
    __shared__ uint32_t hist_all[NW][256];
    __shared__ __attribute__((aligned(2048))) uint32_t own_ab[NW][8 * WAVE];   // a wave at a group boundary: its private partial
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    uint32_t *hist = hist_all[w];
    for (int i = lane; i < 256; i += WAVE) hist[i] = 0;
    const uint32_t T = (uint32_t)d.T;
    const int last = d.ntiles - 1;
    const int64_t nblk = (Q + NW - 1) / NW;
    const uint8_t *rows = d.rows;
    const uint8_t *nib = d.nib;
    const uint8_t *planes = reinterpret_cast<const uint8_t *>(d.planes);
    const uint8_t *planes3 = reinterpret_cast<const uint8_t *>(d.planes3);
    const uint32_t pitch = (uint32_t)d.pitch, npitch = pitch / 2, wpitch = (uint32_t)d.W * 16u,
                   w3pitch = (uint32_t)d.W * 12u;
    constexpr int NJOB = 2;
    __syncthreads();

    const int64_t nslot = xcd_chunk > 0 ? 8 * xcd_chunk : nblk;
    for (int64_t blk0 = blockIdx.x; blk0 < nslot; blk0 += gridDim.x) {
        int64_t blk = blk0;
        if (xcd_chunk > 0) {
            const int64_t x = blk0 & 7, j = blk0 >> 3;
            blk = x * xcd_chunk + j;
            if (j >= xcd_chunk || blk >= nblk) continue;       // uniform for the whole workgroup
        }
        const int64_t it0 = blk * NW;
        const int64_t lqi = order ? (int64_t)order[it0] : it0;
        const uint4 lq = reinterpret_cast<const uint4 *>(quartets)[lqi];
        uint32_t la = __builtin_amdgcn_readfirstlane(lq.x), lb = __builtin_amdgcn_readfirstlane(lq.y);
        const bool leader_ok = (la < T) & (lb < T);
        if (!leader_ok) la = lb = 0;
        const int64_t it = it0 + w;
        const bool have = it < Q;
        const int64_t qi = have ? (order ? (int64_t)order[it] : it) : 0;
        const uint4 qv = reinterpret_cast<const uint4 *>(quartets)[qi];
        uint32_t q[4];
        q[0] = __builtin_amdgcn_readfirstlane(qv.x);
        q[1] = __builtin_amdgcn_readfirstlane(qv.y);
        q[2] = __builtin_amdgcn_readfirstlane(qv.z);
        q[3] = __builtin_amdgcn_readfirstlane(qv.w);
        const bool bad = (q[0] >= T) | (q[1] >= T) | (q[2] >= T) | (q[3] >= T);
        const bool work = have && !bad;
        const bool shares = work && leader_ok && q[0] == la && q[1] == lb;
        const uint32_t qc = work ? q[2] : 0, qd = work ? q[3] : 0;
        const uint32_t l16 = (uint32_t)lane * 16u, l12 = (uint32_t)lane * 12u;
        const uint32_t opc = qc * w3pitch + l12, opd = qd * w3pitch + l12;
        auto fetch_x = [=](int job, int tile) -> uint4 {
            if (job == 0) return ld16(nib, la * npitch + l16 + (uint32_t)tile * (TILE / 2));
            if (job == 1) return ld16(planes, la * wpitch + l16 + (uint32_t)tile * (WAVE * 16));
            return make_uint4(0, 0, 0, 0);
        };
        auto fetch_y = [=](int job, int tile) -> uint4 {
            if (job == 0) return ld16(nib, lb * npitch + l16 + (uint32_t)tile * (TILE / 2));
            if (job == 1) return ld16(planes, lb * wpitch + l16 + (uint32_t)tile * (WAVE * 16));
            return make_uint4(0, 0, 0, 0);
        };
        // eight dwords of one lane into rows 0-7 of a transposed block (row j = 64 consecutive dwords: ds_write_addtid_b32)
        auto store_t = [=](uint32_t *blk, const uint32_t (&v)[8]) {
#ifndef TQ_NO_ASM
            const uint32_t base = __builtin_amdgcn_readfirstlane(lds_offset(blk));
            asm volatile("s_mov_b32 m0, %[base]\n\t"
                         "s_nop 0\n\t"
                         "ds_write_addtid_b32 %[p0]\n\t"
                         "ds_write_addtid_b32 %[p1] offset:256\n\t"
                         "ds_write_addtid_b32 %[p2] offset:512\n\t"
                         "ds_write_addtid_b32 %[p3] offset:768\n\t"
                         "ds_write_addtid_b32 %[p4] offset:1024\n\t"
                         "ds_write_addtid_b32 %[p5] offset:1280\n\t"
                         "ds_write_addtid_b32 %[p6] offset:1536\n\t"
                         "ds_write_addtid_b32 %[p7] offset:1792"
                         :
                         : [base] "s"(base), [p0] "v"(v[0]), [p1] "v"(v[1]), [p2] "v"(v[2]), [p3] "v"(v[3]), [p4] "v"(v[4]),
                           [p5] "v"(v[5]), [p6] "v"(v[6]), [p7] "v"(v[7])
                         : "memory", "m0");
#else
#pragma unroll
            for (int j = 0; j < 8; ++j) blk[j * WAVE + lane] = v[j];
#endif
        };
        auto publish = [=](int b, int job, uint4 x, uint4 y) {
            if (job == 0) {
                const uint32_t h = 0xF0F0F0F0u;
                const uint32_t s0 = (x.x << 2) + y.x, s1 = (x.y << 2) + y.y, s2 = (x.z << 2) + y.z, s3 = (x.w << 2) + y.w;
                const uint32_t v[8] = {(s0 << 4) & h, s0 & h, (s1 << 4) & h, s1 & h, (s2 << 4) & h, s2 & h, (s3 << 4) & h, s3 & h};
                store_t(abp_t[b], v);
            } else if (job == 1) {
                r1_img[b][lane] = make_uint4(x.y, x.z, x.x | y.x, (x.y ^ y.y) | (x.z ^ y.z));
                b_img[b][lane] = x.w;
            }
        };

        auto run = [&](auto spec_tag, auto fast_tag) {
            constexpr int SPEC = decltype(spec_tag)::value;
            constexpr bool FAST = decltype(fast_tag)::value;
            const int job = SPEC >= 0 ? SPEC : w;
            uint4 sx = fetch_x(job, 0), sy = fetch_y(job, 0);
            u32x3 pc = ldv12(planes3, opc), pd = ldv12(planes3, opd);
            publish(0, job, sx, sy);
            uint32_t tile_carry = 0;
            __syncthreads();
            // one step on the plane records (cpc, cpd); those of step t+1 go into (npc, npd) at its top -- the walk reads
            // the current ones -- and the loop below alternates the two register pairs (unrolled by two: no copies)
            auto one_step = [&](const u32x3 &cpc, const u32x3 &cpd, u32x3 &npc, u32x3 &npd, int t) {
                const int tn = min(t + 1, last);
                sx = fetch_x(job, tn);
                sy = fetch_y(job, tn);
                npc = ldv12(planes3, opc + (uint32_t)tn * (WAVE * 12));
                npd = ldv12(planes3, opd + (uint32_t)tn * (WAVE * 12));
                __builtin_amdgcn_sched_barrier(0);
                if (FAST || work) {
                    uint4 r1;
                    uint32_t Bw;
                    const uint8_t *img;
                    if (FAST || shares) {
                        r1 = r1_img[t & 1][lane];
                        Bw = b_img[t & 1][lane];
                        img = reinterpret_cast<const uint8_t *>(abp_t[t & 1] + lane);
                    } else {                                    // group boundary: private rows a and b
                        const uint32_t o0 = q[0] * pitch + (uint32_t)t * TILE + l16;
                        const uint32_t o1 = q[1] * pitch + (uint32_t)t * TILE + l16;
                        const uint4 a0 = ld16(rows, o0), a1 = ld16(rows, o0 + 1024u);
                        const uint4 b0 = ld16(rows, o1), b1 = ld16(rows, o1 + 1024u);
                        const uint32_t v[8] = {((a0.x << 2) + b0.x) << 4, ((a0.y << 2) + b0.y) << 4, ((a0.z << 2) + b0.z) << 4,
                                               ((a0.w << 2) + b0.w) << 4, ((a1.x << 2) + b1.x) << 4, ((a1.y << 2) + b1.y) << 4,
                                               ((a1.z << 2) + b1.z) << 4, ((a1.w << 2) + b1.w) << 4};
                        store_t(own_ab[w], v);
                        const uint4 pa = ld16(planes, q[0] * wpitch + (uint32_t)t * (WAVE * 16) + l16);
                        const uint4 pb = ld16(planes, q[1] * wpitch + (uint32_t)t * (WAVE * 16) + l16);
                        r1 = make_uint4(pa.y, pa.z, pa.x | pb.x, (pa.y ^ pb.y) | (pa.z ^ pb.z));
                        Bw = pa.w;
                        img = reinterpret_cast<const uint8_t *>(own_ab[w] + lane);
                    }
                    const uint32_t C = count_mask_shared<SUB>(r1, Bw, cpc, cpd, lane, tile_carry);
                    walk_planes(C, cpc.y, cpc.z, cpd.y, cpd.z, img, hist);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (job < NJOB) {
                    pin4(sx);
                    pin4(sy);
                }
                publish((t + 1) & 1, job, sx, sy);
                __syncthreads();
            };
            u32x3 qc_, qd_;
            for (int t = 0; t < d.ntiles; t += 2) {
                one_step(pc, pd, qc_, qd_, t);
                if (t + 1 >= d.ntiles) break;
                one_step(qc_, qd_, pc, pd, t + 1);
            }
        };
        {
            using std::integral_constant;
            if (shares) {
                if (w == 0) run(integral_constant<int, 0>{}, integral_constant<bool, true>{});
                else if (w == 1) run(integral_constant<int, 1>{}, integral_constant<bool, true>{});
                else run(integral_constant<int, NJOB>{}, integral_constant<bool, true>{});
            } else {
                run(integral_constant<int, -1>{}, integral_constant<bool, false>{});
            }
        }
        if (have) {
            uint32_t *out = cm + qi * 256;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int bin = lane + WAVE * k;
                __builtin_nontemporal_store(work ? hist[bin] : 0u, &out[bin]);
                hist[bin] = 0;
            }
        }
        __syncthreads();
    }
}

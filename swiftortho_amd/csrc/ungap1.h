// ungap1.h -- device code shared by the ungapped-extension kernels of k_ungap1.hip (a wave per bucket of a bucketed pass: singletons,
// chains) and k_ungapq.hip (a wave per query of a sparse pass): the lane-private score table, the 16-element two-direction X-drop
// step, the pass-record layout and the slot reservation of the pass list.  See the head of k_ungap1.hip.
#pragma once
#include "common.h"
#include "kernels.h"

#define U1_WAVES 16
#define U1_QPAD 32                       // sentinel bytes on both sides of the query in its LDS slot
#define U1_RING 128                      // ring of decoded singletons per wave (power of two)
#define U1_PCAP 32                       // buffered passing groups per wave
#define U1_ROWS 25                       // query classes 0..23 + the sentinel row
#define U1_SENT 24
#define U1_PIN (-8192)
#define U1_CHUNK 512                      // pass-list slots a wave reserves at a time
#define U1_LCHUNK 1024                    // chain-list slots a wave reserves at a time
// per wave: query slot, ring (subject offset, hit word); the buffered passing singletons (hit word | score << 32) of wave w sit in the
// unused tail of table row w when the entries are 16-bit (columns 25..31 of a row: 448 bytes), else behind the ring
#define U1_WAVE_BYTES(QCAP, TSH) ((QCAP) + 2 * U1_QPAD + U1_RING * 8 + ((TSH) == 3 ? 0 : U1_PCAP * 8))

typedef short pk16 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u16 u1_lds_u16;

__device__ __forceinline__ void u1_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint4 u1_load16(const u8* p) {   // unaligned 16-byte GLOBAL load (global_load_dwordx4)
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
// 16 bytes at an arbitrary byte offset of the (16-byte aligned) query slot.  A misaligned LDS read of any width is served one lane per
// cycle (64 cycles per wave-instruction, tools/ubench/ldsua.hip: two of them per step cost as much as the step's 190 vector
// instructions), so: five aligned dwords and four v_alignbyte.
__device__ __forceinline__ uint4 u1_lds16(const u8* slot, int off) {
    const u32* p = reinterpret_cast<const u32*>(slot + (off & ~3));
    const u32 d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
    const u32 sh = (u32)off & 3u;
    return make_uint4(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh), __builtin_amdgcn_alignbyte(d3, d2, sh),
                      __builtin_amdgcn_alignbyte(d4, d3, sh));
}

// ---- score table in LDS: entry (q, s, l) at (q << (8 + TSH)) | (s << (3 + TSH)) | l * (TSH == 4 ? 4 : 2) ----
// TSH: log2(bytes per (query class, subject class * 8) unit) - 3: 4 = 32-bit entries (128 B per class pair: every lane of a 32-lane
// LDS group its own bank), 3 = 16-bit entries (two lanes per bank).
template <int TSH>
__device__ __forceinline__ void u1_fill_table(unsigned char* smem, const signed char* __restrict__ b62g) {
    for (u32 i = threadIdx.x; i < U1_ROWS * 32u * 32u; i += 64 * U1_WAVES) {
        const u32 q = i >> 10, s = (i >> 5) & 31u, l = i & 31u;
        const int v = (q < SCLS_N && s < SCLS_N) ? (int)b62g[q * SCLS_N + s] : -100;
        if (TSH == 4) *reinterpret_cast<int*>(smem + ((q << 12) | (s << 7) | (l << 2))) = v & 0xFFFF;
        else *reinterpret_cast<short*>(smem + ((q << 11) | (s << 6) | (l << 1))) = (short)v;
    }
}

// chain state of a right pass: the group of (up to three) elements in which its maximum last rose, with the running scores there
struct U1Track {
    pk16 Mprev, sv1, sv2, sv3;
    int gsel;
};

// COUNT instances: the b62 lookups the reference makes (Fasta.ungap's `flag`, fsearch.py:2467, 2482), recomputed per direction in plain
// 32-bit arithmetic beside the packed passes: an element counts when its pass is alive and it is no sentinel (the reference's loop
// condition ends the pass in front of a sentinel position; the element that drops a pass below the X-drop line is still counted)
struct U1Count {
    int sR, mR, sL, mL;
    bool aR, aL;
    u32 n;
    __device__ __forceinline__ void start() { sR = mR = sL = mL = 0, aR = aL = true; }
};

// ---- one step: 16 elements of the right pass (low halves) and 16 of the left pass (high halves) ----
// qr4 / sr4: the right windows (element k = byte k), ql4 / sl4: the left windows (element k = byte 15 - k).  Returns the drop mask: all
// ones in the halves whose pass has ended.
template <int TSH, bool CHAIN, bool COUNT>
__device__ __forceinline__ u32 u1_step(const uint4& qr4, const uint4& ql4, const uint4& sr4, const uint4& sl4, u32 lanebase, pk16& S, pk16& M, U1Track& tr, int eb,
                                       U1Count& ct) {
    const pk16 c30 = {30, 30}, pinv = {U1_PIN, U1_PIN};
    const u32* qrd = reinterpret_cast<const u32*>(&qr4);
    const u32* qld = reinterpret_cast<const u32*>(&ql4);
    const u32* srd = reinterpret_cast<const u32*>(&sr4);
    const u32* sld = reinterpret_cast<const u32*>(&sl4);
    u32 msk = 0;
    pk16 Smin = {0, 0}, Sa = {0, 0}, Sb = {0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int kl_ = 15 - k;
        const u32 xr = __builtin_amdgcn_perm(qrd[k >> 2], srd[k >> 2], 0x0c0c0400u + (u32)(k & 3) * 0x0101u);
        const u32 xl = __builtin_amdgcn_perm(qld[kl_ >> 2], sld[kl_ >> 2], 0x0c0c0400u + (u32)(kl_ & 3) * 0x0101u);
        const u32 ar = (xr << TSH) + lanebase, al = (xl << TSH) + lanebase;
        const u32 cr = *(u1_lds_u16*)(size_t)ar, cl = *(u1_lds_u16*)(size_t)al;
        const pk16 C = __builtin_bit_cast(pk16, __builtin_amdgcn_perm(cl, cr, 0x05040100u));
        if (COUNT) {
            const int vr = (short)cr, vl = (short)cl;
            if (ct.aR) {
                if (vr == -100) ct.aR = false;
                else {
                    ++ct.n, ct.sR += vr, ct.mR = max(ct.mR, ct.sR);
                    if (ct.mR - ct.sR > DROPX) ct.aR = false;
                }
            }
            if (ct.aL) {
                if (vl == -100) ct.aL = false;
                else {
                    ++ct.n, ct.sL += vl, ct.mL = max(ct.mL, ct.sL);
                    if (ct.mL - ct.sL > DROPX) ct.aL = false;
                }
            }
        }
        S += C;
        M = __builtin_elementwise_max(M, S);
        // X-drop, tested once per group of <= 3 elements (k = 2, 5, 8, 11, 14, 15) on the group's LOWEST running score against the
        // maximum at the group's end: a pass dropped inside the group iff that minimum is <= maximum - 31.  Exact: a maximum
        // that rose inside the group rose before any drop (after a drop the score cannot exceed it within two elements), and
        // a score 31 below the final maximum cannot have climbed to it within two elements either (2 * 11 < 31).
        Smin = (k % 3 == 0) ? S : __builtin_elementwise_min(Smin, S);
        if (CHAIN) {
            if (k % 3 == 0) Sa = S;
            if (k % 3 == 1) Sb = S;
        }
        if (k % 3 == 2 || k == 15) {
            if (CHAIN) {   // the group of elements in which the right pass's maximum last rose, with its running scores
                const bool chg = M.x != tr.Mprev.x;
                tr.gsel = chg ? eb + (k - k % 3) : tr.gsel;
                tr.sv1 = chg ? Sa : tr.sv1, tr.sv2 = chg ? (k == 15 ? S : Sb) : tr.sv2, tr.sv3 = chg ? S : tr.sv3;
                tr.Mprev = M;
            }
            // all ones in the halves whose pass has dropped (now or earlier: a dropped pass is pinned at -8192).  (The empty asm
            // statements keep this as sub + sub + shift + one bit-select; left alone the compiler goes through two 16-bit
            // compares, two selects and a v_perm.)
            u32 t30 = __builtin_bit_cast(u32, c30 - (M - Smin));
            asm volatile("" : "+v"(t30));
            msk = __builtin_bit_cast(u32, __builtin_bit_cast(pk16, t30) >> 15);
            asm volatile("" : "+v"(msk));
            S = __builtin_bit_cast(pk16, (__builtin_bit_cast(u32, S) & ~msk) | (__builtin_bit_cast(u32, pinv) & msk));
        }
    }
    return msk;
}

// pass record of a group with head word fw in bucket (range r, batch query gq): p_qs = (q << bs) | subject, p_sd = (score << 32) |
// (sst - qst), p_ft = the head hit's key in the record layout (k_rec_scatter turns it into the first-touch key)
template <bool BANDS>
__device__ __forceinline__ void u1_record(u32 fw, u64 score_hi, u32 r, u32 gq, const BktLayout& L, int diag_off, int rbs, int rsh_subj, int rsh_diag, int rdoff,
                                          int sh_qpos, const uint2* __restrict__ btab, u64* __restrict__ p_qs, u64* __restrict__ p_sd, u64* __restrict__ p_ft, u32 at) {
    const u32 G = (r << (L.wb + L.bd)) | (fw >> L.bp);
    u32 gsubj;
    int dlt;
    if (BANDS) {
        const uint2 be = btab[G >> L.bd];
        gsubj = be.x;
        dlt = (int)(be.y - G);
    } else {
        gsubj = G >> L.bd;
        dlt = diag_off - (int)(G & ((1u << L.bd) - 1u));
    }
    p_qs[at] = ((u64)gq << rbs) | gsubj;
    p_sd[at] = score_hi | (u64)(u32)dlt;
    p_ft[at] = ((u64)gsubj << rsh_subj) | ((u64)(u32)(rdoff - dlt) << rsh_diag) | ((u64)(fw & ((1u << L.bp) - 1u)) << sh_qpos);
}

// Slots of the pass list come from the wave's reserved piece; a new piece of U1_CHUNK slots costs ONE atomic.  (One atomic per flush of
// <= 64 records was 1-2 million same-address atomics per launch, served at ~10 ns each: as long as the kernel itself.)  Returns the slot
// of lane `lane`'s record (lane < n).
__device__ __forceinline__ u32 u1_reserve(u32 n, int lane, u32& ch_pos, u32& ch_end, u32* __restrict__ counter, u32 chunk) {
    const u32 avail = ch_end - ch_pos;
    u32 nbase = 0;
    if (n > avail) {
        if (lane == 0) nbase = atomicAdd(counter, chunk);
        nbase = (u32)__builtin_amdgcn_readfirstlane((int)nbase);
    }
    const u32 at = (u32)lane < avail ? ch_pos + (u32)lane : nbase + ((u32)lane - avail);
    if (n > avail) ch_pos = nbase + (n - avail), ch_end = nbase + chunk;
    else ch_pos += n;
    return at;
}


/*
 * gkm_gram_bitslice.hip -- the HOT kernel of the gkm kernel-matrix path on MI355X (gfx950): bit-sliced diagonal mismatch
 * profiles -> raw Gram values, tile-transposed (replaces the k-mer tree DFS of src/libgkm.c:315-387
 * and the per-row reduction of :553-589).  Written for CDNA4 only: 64-wide wavefronts, one wavefront per workgroup, the
 * column tables streamed through the scalar unit (SGPRs), a per-wave hit list in LDS.  The launch geometry (row packing,
 * work-item order, tables) is gkm_gram.hip's; the per-lane arithmetic is gkm_bitslice.h's (unit-tested on the CPU).
 */
#include "gkm_gram_bitslice.h"

/* wave64 inclusive prefix sum on the DPP network (no LDS round trips): four row_shr steps
 * scan each row of 16 lanes, row_bcast:15 / row_bcast:31 carry the row totals across */
__device__ __forceinline__ int wave_inclusive_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); /* row_shr:1 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); /* row_shr:2 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); /* row_shr:4 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); /* row_shr:8 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); /* row_bcast:15 -> rows 1,3 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); /* row_bcast:31 -> rows 2,3 */
    return x;
}

/* position of the lowest set bit, 0xFFFFFFFF for 0 (v_ffbl_b32's own convention; __builtin_ctz(0) is
 * undefined and the generic cttz costs a second instruction) */
__device__ __forceinline__ uint32_t ffbl_or_ones(uint32_t x)
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

/* 2 x as an addition: on gfx950 v_lshlrev_b32 issues at HALF the rate of v_add_u32 (tools/valu_ops.hip), and hipcc
 * turns x + x back into a shift */
__device__ __forceinline__ uint32_t twice(uint32_t x)
{
    uint32_t r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}

/* |a - b| in one instruction (hipcc expands __usad(a, b, 0) into v_min / v_max / v_sub) */
__device__ __forceinline__ uint32_t absdiff(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

/* popcount(x) + acc in one instruction (hipcc sums separate popcounts with extra adds) */
__device__ __forceinline__ uint32_t popc_add(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

#ifndef GKM_BS_GRP
#define GKM_BS_GRP 5 /* config 2 / gkmQC's default L=10 k=6 d=3: 2 -> 89.8 / 118.1 ms, 5 -> 81.0 / 119.3, 10 -> 87.5 / 138.3 */
#endif
constexpr int BS_GRP = GKM_BS_GRP;     /* hit words per list record: the lanes are compacted once per BS_GRP words */
#ifndef GKM_BS_TRIP
#define GKM_BS_TRIP 64 /* config 2: 64 -> 87.2 ms (ring of 128: index wrap is one AND), 128 -> 89.1, 192 -> 97.3 */
#endif
constexpr int BS_TRIP = GKM_BS_TRIP; /* records resolved per trip: one per lane */
static_assert(BS_TRIP == 64, "a trip resolves one record per lane");
/* records the wave-wide hit list holds: >= BS_TRIP + 64, a multiple of 64 (merged LDS stores) */
constexpr int BS_CAP = BS_TRIP + 64;
#ifndef GKM_BS_WAVES
#define GKM_BS_WAVES 7 /* waves per SIMD asked of the compiler for the same-length kernel, PK = 4 (<= 72 VGPRs).  Round 1, with
                          the grouped hit ring: 5 -> 96.1 ms, 6 -> 91.8, 7 (spills) -> 97.5 on config 2; rounds 2-4 compiled for 6 and
                          came out at 69-72 VGPRs, i.e. ran 7.  Round 5's second hit of a visit takes 73 when compiled for 6;
                          compiled for 7 it fits 72 without scratch: config 2 69.7 -> 68.6 ms, gkmQC's shape 357.3 -> 349.1
                          (profiles/r5_kernel_ab_hit_path.txt) */
#endif
#ifndef GKM_BS_PACKED_WAVES
#define GKM_BS_PACKED_WAVES 6 /* every other variant (ragged lengths): 71-78 VGPRs; compiled for 7 the one-piece ones spill */
#endif
#ifndef GKM_TRIP_PRIO
#define GKM_TRIP_PRIO 3 /* wave priority (s_setprio, 0..3) inside a trip; 0 = as rounds 1-3 */
#endif

/*
 * One wavefront = 64 row segments (one per lane) x ONE column sequence.
 * For both strands of the column the wave sweeps all T cyclic shifts; per shift each lane
 * evaluates 32*W l-mer window comparisons with ~13 VALU instructions per 32 (gkm_bitslice.h).
 * Hit words are parked, compacted over the lanes, in a wave-wide LDS list (a stack) of records and turned
 * into weighted profile counts in full-wave batches, so the hot loop has no data-dependent
 * control flow besides the push.
 */
template <int W, int L, int D, int PK>
__global__ __launch_bounds__(64, D > 4 ? 1 : PK == 4 ? GKM_BS_WAVES : GKM_BS_PACKED_WAVES) void k_gram_bitslice(const BsArgs A)
{
    /* PK = 4: problems whose sequences all have the same length (gkmQC's own 600-bp subsets, BASELINE configs 1-3);
     *      1: everything else -- several pieces per lane (gkm_pack.h), up to 64 rows per tile; 2: up to 128 rows per tile;
     *      0: one piece per lane, ragged lengths, piece entries in a 512-byte LDS table; 3: the same with the entries
     *         fetched from the source lane's registers (ds_bpermute_b32) where those 512 bytes cost an LDS allocation
     *         granule.  Since round 5 only for L < 5 and in the tests (GKM_NO_UNIF): with L >= 5 the variants 4, 1, 2
     *         resolve hits by GROUPS (below), which beats both. */
    constexpr bool PACKED = PK == 1 || PK == 2;
    constexpr bool BPERM = PK == 3;
    /*      4: a row takes k = ceil(windows / 310) whole lanes, piece pi of it starts at sequence position pi * capacity,
     *         and all a trip needs of the source lane -- its row slot and pi -- rides in the record's origin word (9 spare
     *         bits, set once per wave): no piece table, no permute, and the row l-mer's weight comes from the column's own
     *         table by position (same length, same weights).  Round 5: an LDS operation in a trip costs what three VALU
     *         instructions do (sensitivity probes, profiles/r5_trip_sensitivity.txt). */
    constexpr bool UNIF = PK == 4;
    /* (The ablation builds of rounds 1-3 -- parts of this kernel skipped to time the rest, results wrong -- lived
     * here as a fifth template parameter; they are gone from the source since round 4.  tools/variants.sh rebuilds
     * them from revision a4bed73, profiles/r2_ablation_timings.txt and r2_pmc_ablation_builds*.txt hold what they
     * measured.) */
    using namespace gkmbs;
    /* LDS per wave.  STATIC, one array carved by hand so that the mismatch profiles come FIRST (see `resolve`: a hit adds
     * at accl + m * NSLOT + slot without testing m <= D; the windows with a larger m are the ones that wrap around the
     * end of the column strand, they carry the weight 0, and wherever m <= L lands it is inside this array):
     *   accl   [(D + 1) * NSLOT]     mismatch profiles [m][row slot]                              1-2.5 KB
     *   s_list [2 | BS_GRP + 1][CAP]  the hit list: two-word group records (GROUP), else five hit words + origin   1 | 3 KB
     *   lmask  [64]                  PACKED without group records: piece-start bit rows of every lane   0.25 KB
     *   lpiece [64 * NP | 128 | 0]   piece entries (none in the BPERM variant)                    0-1 KB
     * DYNAMIC: the column's two 2-bit packed strands, interleaved word by word (2 * pkw words: 0.2 KB at 300 bp, 0.3 KB
     * at 600 bp), then the weight tables: the column's weights by l-mer position (gkm_gram_bitslice.h POSTAB_PAD; ~T + L
     * + 16 bytes), behind it the row side's -- none in the same-length variant (the rows read the column's table), the
     * centred distance table in the several-pieces group variants, a copy of the distance table otherwise.  What a visit
     * reads: two words of the column strand and the weight bytes from LDS; two words of the row lane's packed positions
     * (8 KB per tile -- 128 bytes per lane, of which 84 are used -- L1 resident: the waves of a CU work on the same
     * tile) from global memory. */
    extern __shared__ uint32_t s_dyn[];
    /* PACKED: lanes may hold several pieces (gkm_pack.h).  When no lane of the call holds more than one
     * piece (e.g. every fixed-length data set) the leaner variant runs: one (slot, centre) pair per lane.
     * The several-pieces variant exists for 64 and for 128 row slots per tile: the profiles of 128 slots
     * (2.5 KB at d = 4) cost a wave per SIMD, so the host packs at most 64 rows into a tile unless
     * that would leave lanes empty (many rows shorter than half a lane). */
    constexpr int NP = PACKED ? gkmpack::MAX_PIECES : 1;   /* pieces per lane */
    constexpr int NSLOT = PK == 2 ? gkmpack::MAX_ROWS : 64; /* row slots per tile */
    /* GROUP (the same-length variant): a record of the hit list is TWO words -- the OR of the group's five hit words
     * (which bit rows of the lane hold a hit somewhere in the group) and the origin -- and a trip finds the hits among the
     * five windows of (bit row, group) itself, from the packed strands it reads anyway (see trip_group). */
    constexpr bool GROUP = UNIF || (PACKED && L >= 5); /* (L >= 5: the L - 1 zero bytes of a weight table cover a group) */
    constexpr bool PGROUP = GROUP && PACKED;
    constexpr int LIST_ARRAYS = GROUP ? 2 : BS_GRP + 1;
    constexpr int ACC_WORDS = (D + 1) * NSLOT, LIST_WORDS = LIST_ARRAYS * BS_CAP, LMASK_WORDS = (PACKED && !PGROUP) ? 64 : 0;
    /* per piece: row slot * 4 and the biased centre offset c0 + 2048 -- two words in the one-piece variant
     * (one ds_read_b64), one word (slot * 4 | c0b << 16) in the several-pieces variants */
    constexpr int LPIECE_WORDS = PACKED ? 64 * NP : (BPERM || UNIF) ? 0 : 128;
    constexpr int STATIC_WORDS = ACC_WORDS + LIST_WORDS + LMASK_WORDS + LPIECE_WORDS;
    __shared__ uint32_t s_mem[STATIC_WORDS];
    uint32_t *const accl = s_mem; /* mismatch profiles [m][row slot] */
    /* The hit list.  A record is the BS_GRP hit words of one lane for BS_GRP consecutive words of a
     * shift plus their origin; word k of record i sits at s_list[k * BS_CAP + i], the origin (first word of the
     * group, shift, row lane) at k = BS_GRP (arrays a multiple of 64 dwords apart: the stores of a push merge into
     * ds_write2st64_b32).  Compacting once per group instead of once per word takes 3 VALU
     * instructions per word out of the hot loop (config 2: 111.0 -> 96.2 ms). */
    uint32_t *const s_list = s_mem + ACC_WORDS;
    uint32_t *const lmask = s_mem + ACC_WORDS + LIST_WORDS;
    uint32_t *const lpiece = lmask + LMASK_WORDS;
    static_assert((ACC_WORDS * 4) % 256 == 0, "the list's arrays stay 64-dword aligned (ds_write2st64_b32)");
    /* an add at accl[m <= L][slot] stays inside the static LDS (everywhere but 128 slots with d <= 1) */
    constexpr bool M_FITS = (L + 1) * NSLOT <= STATIC_WORDS;
    /* (the dynamic LDS follows the static LDS: the column image's address is STATIC_WORDS * 4) */
    constexpr bool COL_BASE_FOLDS = (STATIC_WORDS * 4) % 1024 == 0;
    static_assert(W % BS_GRP == 0, "a shift is a whole number of record groups");
    /* The list is a STACK (round 3; a ring before): a trip is due as soon as it holds BS_TRIP records and it is checked
     * after every group (at most 64 new records); a trip takes the BS_TRIP records on TOP and puts at most as many back:
     * the list never holds more than BS_TRIP + 63 records.  The order in which hits are resolved is immaterial
     * (integer adds), and a stack needs no head and no wrap: one AND less per push, per trip and per re-push, and the
     * trip's read address is lane * 4 + a scalar. */
    static_assert(BS_CAP >= BS_TRIP + 64 && BS_CAP % 64 == 0, "hit list too small");
    static_assert(gkmpack::MAX_ROWS % 64 == 0, "row slots are finished 64 at a time");

    const int lane = threadIdx.x;
    /* (raising the priority of a NEW wave too, until its row planes are loaded, was measured: 395.4 against 388.9 ms on
     * gkmQC's shape, nothing on config 2 -- profiles/r4_kernel_ab_trip_priority.txt) */
    /* block -> (tile, column): see BsArgs.  All of this is wave-uniform (scalar loads, SALU). */
    int tile, j0;
    if (A.nent > 0) { /* (chunk, tile) entries: largest e with ent_off[e] <= blockIdx.x */
        int lo = 0, hi = A.nent;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (A.ent_off[mid] <= (int64_t)blockIdx.x) lo = mid;
            else hi = mid;
        }
        tile = A.ent_tile[lo];
        j0 = A.ent_j0[lo] + (int)((int64_t)blockIdx.x - A.ent_off[lo]);
        if (j0 >= A.ent_j1[lo]) return; /* padding item */
    } else {
        int lo = 0, hi = A.ntiles; /* largest tile with tile_soff[tile] <= blockIdx.x */
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (A.tile_soff[mid] <= (int64_t)blockIdx.x) lo = mid;
            else hi = mid;
        }
        tile = lo;
        j0 = A.tile_cbeg[tile] + (int)((int64_t)blockIdx.x - A.tile_soff[tile]);
    }
    const int j1 = j0 + 1;
    const int nrows = A.tile_nrows[tile];
    constexpr int NE = NSLOT / 64; /* row slots a lane finishes in the epilogue */
    /* A tile with at most NSLOT / 2 rows (600-bp rows: 32 per tile) keeps TWO copies of every profile, NSLOT / 2 slots
     * apart, and the epilogue adds them: the hits of odd source lanes go to the second copy (the host puts the offset
     * into those lanes' piece entries, gram_launch), so the ds_add_u32 of a trip spread over twice the addresses --
     * 88 % of all hits have m = d and 64 lanes were adding into 32 words. */
    const bool two_copies = 2 * nrows <= NSLOT;

    uint32_t Ahi[W], Alo[W], AV[W];
#pragma unroll
    for (int w = 0; w < W; w++) {
        Ahi[w] = A.rowplanes[(((size_t)tile * 3 + 0) * W + w) * 64 + lane];
        Alo[w] = A.rowplanes[(((size_t)tile * 3 + 1) * W + w) * 64 + lane];
        AV[w] = A.rowplanes[(((size_t)tile * 3 + 2) * W + w) * 64 + lane];
    }

    /* one wavefront per workgroup: the LDS traffic of a wave is ordered, no barriers needed */
    /* (PGROUP: the piece-start mask stays in a register of its lane and a trip fetches it by ds_bpermute_b32 -- the 256
     * bytes keep the wave inside four LDS allocation granules at 600 bp) */
    uint32_t my_lmask = 0u;
    if (PGROUP) my_lmask = A.lane_mask[tile * 64 + lane];
    else if (PACKED) lmask[lane] = A.lane_mask[tile * 64 + lane];
    constexpr int LPW = PACKED ? NP : 2; /* lpiece words per lane */
    if (!BPERM && !UNIF) {
#pragma unroll
        for (int k = 0; k < LPW; k++) lpiece[lane * LPW + k] = A.lane_piece[(size_t)(tile * 64 + lane) * LPW + k];
    }
    /* BPERM: the lane keeps its own (row slot * 4, biased centre offset) in two registers and a trip fetches
     * the source lane's pair over the permute network (ds_bpermute_b32: no LDS storage, no bank conflicts).
     * The 512 bytes this takes out of LDS bring a wave under 5 120 bytes = 4 allocation granules of 1 280
     * (tools/lds_occupancy.hip): 32 instead of 24 one-wave workgroups fit a CU at 600 bp. */
    uint32_t my_both = 0u; /* row slot * 4 (< 256) | biased centre offset (< 8192) << 16 */
    if (BPERM)
        my_both = A.lane_piece[(size_t)(tile * 64 + lane) * 2] | (A.lane_piece[(size_t)(tile * 64 + lane) * 2 + 1] << 16);
    /* (UNIF: word 0 of the lane's piece entry holds its row slot and piece index where the origin word wants them) */
    const uint32_t lane_tag = ((uint32_t)lane << META_LANE_SHIFT) | (UNIF ? A.lane_piece[(size_t)(tile * 64 + lane) * 2] : 0u);
    const uint32_t lane4 = (uint32_t)lane << 2;
    const int pkw = A.pkw;
    /* dynamic LDS: the column's two packed strands first, interleaved word by word, the weight bytes behind them */
    uint32_t *const s_col = s_dyn;
    uint8_t *const s_wtab = (uint8_t *)(s_dyn + 2 * pkw);
    /* POSTAB (the one-piece-per-lane variants): the COLUMN's weights sit in LDS by l-mer position with L - 1 zero bytes
     * either side, which rids a trip of its two tests (below); the several-pieces variants keep the distance-indexed
     * table for both sides and the tests -- their LDS has no room for T + L - 1 more bytes without losing a wave. */
    constexpr bool POSTAB = !PACKED;
    /* this tile's packed lanes: 32-bit byte offsets from a wave-uniform base (global_load with an SGPR
     * base instead of a 64-bit address computed per lane); 128 bytes per lane, so that the lane field of a
     * record's origin word IS the lane's byte offset */
    const char *const rowpk_tile = (const char *)(A.rowpk + (size_t)tile * 64 * A.rpw);

    for (int j = j0; j < j1; j++) {
        const int T = A.len[j];
        const int nB = T - L + 1;
        const uint32_t rcpT = mod_magic((uint32_t)T);
        for (int x = lane; x < 2 * pkw; x += 64) s_col[x] = A.colpk[(size_t)j * 2 * pkw + x];
        /* Weights.  wd[D] = weight of an l-mer at distance D from its sequence's centre l-mer (libgkm.c:912-925 depends on
         * nothing else; ones for the unweighted kernel types), 1 KB in global memory.
         * POSTAB: the column's weights BY POSITION, s_wtab[L - 1 + p] = wt[p] = wd[|nB/2 - p|] for the l-mers p < nB,
         * L - 1 zero bytes either side.  A forward-strand window q reads s_wtab[L - 1 + q]; the reverse strand's weights
         * are the forward ones mirrored, wt_rc[q] = wt[nB-1-q] (libgkm.c:924): s_wtab[L - 1 + nB - 1 - q].  The windows
         * that wrap around the end of the strand (q = nB .. T - 1: not l-mers, gkm_bitslice.h window_hits) land in the
         * zero bytes behind / before the table: they add 0 to some profile word and need no test.  (A distance-indexed
         * table cannot do that: |nB/2 - q| of q = nB equals that of q = 0 when nB is even.)  The ROW side: UNIF reads the
         * very same table by position (every sequence has the column's length); the other one-piece variants read
         * wd[|c0 - i0|] from a copy of wd behind it (A.ptw words on).
         * !POSTAB: one distance-indexed table for both sides, as rounds 2-4 had it. */
        const uint32_t ccen = (uint32_t)(nB / 2);
        uint32_t s_rowbase; /* LDS byte offset (from s_dyn) of the row side's wd[0] */
        if (POSTAB || PGROUP) {
            /* (built once per sequence by k_build_postab, gkm_context.hip: the wave copies dwords, as it copies the strands) */
            for (int x = lane; x < A.ptw; x += 64) ((uint32_t *)s_wtab)[x] = A.postab[(size_t)j * A.ptw_stride + x];
            if (PGROUP) {
                /* the row side's weights CENTRED: byte A.wdc_centre + s = wd[|s|] for the signed distance s of an l-mer to
                 * its sequence's centre l-mer, so that five consecutive l-mers read five consecutive bytes (the host builds
                 * it: gkm_gram.hip; it serves every row length) */
                for (int x = lane; x < A.wdc_words; x += 64) ((uint32_t *)(s_wtab + A.ptw * 4))[x] = A.wdc[x];
                /* (the l-mer at lane position i0 has the signed distance i0 + 2048 - c0b: its byte is at i0 - c0b + this) */
                s_rowbase = (uint32_t)(STATIC_WORDS * 4) + (uint32_t)pkw * 8u + (uint32_t)A.ptw * 4u + (uint32_t)A.wdc_centre + 2048u;
            } else {
                if (!UNIF)
                    for (int x = lane; x < A.wd_words; x += 64) ((uint32_t *)(s_wtab + A.ptw * 4))[x] = ((const uint32_t *)A.wd8)[x];
                s_rowbase = (uint32_t)pkw * 8u + (UNIF ? POSTAB_PAD + (uint32_t)(L - 1) /* by position */ : (uint32_t)A.ptw * 4u) +
                            (GROUP ? (uint32_t)(STATIC_WORDS * 4) : 0u); /* (GROUP: an LDS address, not an offset into s_dyn) */
            }
        } else {
            for (int x = lane; x < A.wd_words; x += 64) ((uint32_t *)s_wtab)[x] = ((const uint32_t *)A.wd8)[x];
            s_rowbase = (uint32_t)pkw * 8u;
        }
        /* strand-uniform scalars of the hit path (set at the top of each strand's sweep; the list is emptied between
         * the strands, so a trip only ever holds records of ONE strand and the strand costs it no instruction):
         *   s_strand4   byte offset of the strand's words in the interleaved column image (0 / 4)
         *   POSTAB:  s_wsign 0 / ~0, s_wbase: the column weight's LDS byte is (q ^ s_wsign) + s_wbase = L-1 + q or
         *            L-1 + nB-1 - q behind the table's start
         *   !POSTAB: s_wbase = [reverse strand and nB even]: wt_rc[q] = wt[nB-1-q] = wd[|q + [nB even] - nB/2|] */
        uint32_t s_strand4 = 0u, s_wsign = 0u, s_wbase = 0u, s_wstep = 1u, s_wback = 0u, s_perm4 = 0x03020100u, s_perm1 = 0x0c0c0c04u;
        uint32_t v_rowbase = 0u;
        if (!POSTAB) asm volatile("v_mov_b32 %0, %1" : "=v"(v_rowbase) : "s"(s_rowbase));
#pragma unroll
        for (int m = 0; m <= D; m++)
            for (int rs = lane; rs < (two_copies ? NSLOT : nrows); rs += 64) accl[m * NSLOT + rs] = 0u;
        int s_n = 0; /* records in the hit list (wave-uniform) */

        /* One hit -> accl[m][row slot] += wa * wb.  (meta + sel, bit) name the row lane r, the lane position
         * i0 = bit*W + w of the window, the shift and the strand; lane and bit row name the piece (gkm_pack.h),
         * the piece names the row slot and c0, which makes |c0 - i0| the row l-mer's distance to its sequence's
         * centre l-mer (libgkm.c:912-925 depends on nothing else).  Written for the ISSUE COST -- the kernel is
         * bound by VALU issue, and on gfx950 only the plain two-operand integer operations and v_bitop3_b32 issue
         * at the full rate; v_bfe, v_mad_u32_u24, v_min, v_sad, v_alignbit, v_ffbl, v_bcnt, compares, SDWA and
         * anything with an SGPR operand take twice as long (tools/valu_ops.hip).  Hence the layout of the origin
         * word (gkm_bitslice.h pack_meta: fields that are masked in place or shifted out of the top), 128 bytes per
         * lane of packed positions, the column's strands interleaved word by word, (a & const) | b as one
         * v_bitop3_b32, the strand as wave-uniform scalars (the list is emptied between the strands).
         * Same arithmetic as resolve_hit_packed (gkm_bitslice.h), which the CPU tests run against the oracle. */
        auto resolve = [&](uint32_t ms, uint32_t bit, uint32_t pslot4, uint32_t pc0b, uint32_t cont) -> uint32_t {
            const uint32_t lane128 = ms & (63u << META_LANE_SHIFT); /* source lane * 128 */
            const int k = PACKED ? piece_of_bitrow(*(const uint32_t *)((const char *)lmask + (PACKED ? (lane128 >> 5) : 0u)), (int)bit) : 0;
            uint32_t slot4, c0b = 0u; /* row slot * 4; (l-mers of the row) / 2 - p0 + b0*W + 2048 */
            if (PACKED) {
                static_assert(!PACKED || NP == 4, "lpiece is addressed as lane * 16 + piece * 4");
                const uint32_t lp = *(const uint32_t *)((const char *)lpiece + ((lane128 >> 3) + ((uint32_t)k << 2)));
                slot4 = lp & 0xFFFFu;
                c0b = lp >> 16;
            } else if (UNIF) {
                slot4 = (ms >> (META_SLOT_SHIFT - 2)) & 0xFCu;
            } else if (BPERM) {
                slot4 = pslot4;
                c0b = pc0b;
            } else {
                const uint32_t *lp2 = (const uint32_t *)((const char *)lpiece + (lane128 >> 4));
                slot4 = lp2[0];
                c0b = lp2[1];
            }
            const uint32_t i0 = __umul24(bit, (uint32_t)W) + (ms & 15u);
            const uint32_t x = i0 + (ms >> 21);
            uint32_t q;
            if ((uint32_t)T >= (uint32_t)(32 * W)) q = min(x - (uint32_t)T, x); /* x < 2T (wave-uniform test) */
            else q = mod_small(x, (uint32_t)T, rcpT);
            /* a window that wraps around the end of the strand is not an l-mer (gkm_bitslice.h window_hits); POSTAB: its
             * weight is 0 */
            if (POSTAB || (int)q < nB) {
                /* (a & -4) | b and (a & -8) | b as ONE v_bitop3_b32 each (truth table 0xEA), inline constants */
                const uint32_t *rw = (const uint32_t *)(rowpk_tile + lop3<0xEA>(i0 >> 2, ~3u, lane128));
                /* the column image starts where the static LDS ends; where that is a multiple of 1 024 bytes (d = 3) the
                 * word offset (q / 16 * 8 < 1 024) and the base share no bit and the base rides in the same v_bitop3_b32 */
                typedef const uint32_t __attribute__((address_space(3))) *lds_words; /* (a 32-bit LDS address) */
                const uint32_t cwo = lop3<0xEA>(q >> 1, ~7u, s_strand4);
                const lds_words cw = COL_BASE_FOLDS ? (lds_words)(uintptr_t)cwo : (lds_words)(uintptr_t)((uint32_t)(STATIC_WORDS * 4) + cwo);
                const uint8_t *wdb = (const uint8_t *)s_dyn;
                /* (the table's offset rides in the third operand of the v_sad_u32 that forms the index) */
                uint32_t wa, wb;
                uint32_t ia, ib; /* LDS byte offsets of the two weights */
                if (UNIF) /* the row l-mer is l-mer pi * capacity + i0 of a sequence as long as the column */
                    ia = __umul24((ms >> META_PIECE_SHIFT) & 7u, (uint32_t)segment_capacity(W, L)) + i0 + s_rowbase;
                else if (POSTAB) ia = __usad(c0b, i0 | 2048u, s_rowbase);
                else ia = __usad(c0b, i0 | 2048u, v_rowbase); /* (offset in a VGPR: |q - centre| + offset would name two SGPRs) */
                if (POSTAB) ib = (q ^ s_wsign) + s_wbase;
                else ib = __usad(q + s_wbase, ccen, v_rowbase);
                /* (Round 5 also read the weight bytes in PAIRS -- this window's and its neighbour's, for the second hit of the
                 * visit below, as one 16-bit LDS read each: two LDS operations fewer per trip -- and lost 58 % on gkmQC's
                 * shape (563 against 357 ms, profiles/r5_kernel_ab_hit_path.txt): half of those reads sit at odd addresses,
                 * and whatever the hardware does with a misaligned ds_read_u16, it is no single LDS operation.  Not
                 * pursued with an aligned layout: two bytes per position would cost the LDS allocation granule.) */
                wa = wdb[ia];
                wb = wdb[ib];
                /* (v_alignbit_b32 uses the low 5 bits of its count: 2 i0 mod 32 = 2 (i0 mod 16)) */
                const uint32_t ea = __builtin_amdgcn_alignbit(rw[1], rw[0], twice(i0));
                const uint32_t eb = __builtin_amdgcn_alignbit(cw[2], cw[0], twice(q));
                const uint32_t m = (uint32_t)pk_mismatch(ea, eb, L);
                /* POSTAB: no test for m <= D either.  A window that is an l-mer on both sides has the m the counting loop
                 * found (<= D); one that wraps (packed strands: zeros behind the end, so any m <= L) has wb = 0 and adds
                 * nothing, wherever m * NSLOT + slot lies in the static LDS (M_FITS).  LDS atomic: ds_add_u32. */
                if ((POSTAB && M_FITS) || m <= (uint32_t)D)
                    atomicAdd((uint32_t *)((char *)accl + (m * (uint32_t)(NSLOT * 4) + slot4)), wa * wb);
                /* The NEXT position of the same diagonal (row window i0 + 1 against column window q + 1: the same bit of the
                 * record's next word), when it is a hit too -- 40 % of the visits on gkmQC's shape, because a window pair
                 * within d mismatches makes its neighbour likely (tools/hits_per_record.py).  It shares the record, the
                 * piece and both packed windows: two shifts by 2 bits, a second pair of weight bytes (the neighbours of the
                 * first), a second ds_add. */
                cont &= (uint32_t)(q + 1u != (uint32_t)T); /* (the column window after the strand's last is its first again) */
                if (cont) {
                    uint32_t xr = (ea ^ eb) >> 2;
                    xr = (xr | (xr >> 1)) & (0x55555555u & ((1u << (2 * L)) - 1u));
                    const uint32_t m1 = (uint32_t)__builtin_popcount(xr);
                    uint32_t wa1, wb1;
                    if (UNIF) wa1 = wdb[ia + 1u];
                    else if (POSTAB) wa1 = wdb[__usad(c0b, (i0 | 2048u) + 1u, s_rowbase)];
                    else wa1 = wdb[__usad(c0b, (i0 | 2048u) + 1u, v_rowbase)];
                    if (POSTAB) wb1 = wdb[ib + s_wstep];
                    else wb1 = wdb[__usad(q + 1u + s_wbase, ccen, v_rowbase)];
                    if ((POSTAB && M_FITS) || (m1 <= (uint32_t)D && (int)q + 1 < nB))
                        atomicAdd((uint32_t *)((char *)accl + (m1 * (uint32_t)(NSLOT * 4) + slot4)), wa1 * wb1);
                }
                return cont; /* (the visit took the pair) */
            }
            return 0u;
        };

        /* one trip over the `c` records on top of the list (PARTIAL: c < BS_TRIP, the last trip of a column) */
        auto trip = [&](auto partial_tag, int c) {
            constexpr bool PARTIAL = decltype(partial_tag)::value;
            /* A wave inside a trip issues AHEAD of the waves that are in the counting loop (s_setprio; back to 0 at the
             * end of the trip).  A trip is a chain of short instruction runs between LDS and memory round trips (record
             * -> piece entry -> row words -> column words and weights -> accumulate); at equal priority each run waits
             * its turn behind six waves of straight-line counting code, and the chain -- with the LDS list and the other
             * lanes' hits waiting on it -- stretches.  Round 4, same-run A/B (profiles/r4_kernel_ab_trip_priority.txt):
             * config 2 75.2 -> 72.8 ms, gkmQC's own shape 433.3 -> 396.0 ms, config 5 167.4 -> 152.7 ms; priority 1 and
             * 3 do the same.  The total VALU work is unchanged: this is issue ORDER, not instruction count. */
            __builtin_amdgcn_s_setprio(GKM_TRIP_PRIO);
            /* the c records on top: lane * 4 + a scalar (kept apart from the lane term: hipcc would fuse the shift into a
             * half-rate v_lshl_add_u32 and split the reads around a negative offset) */
            const uint32_t top4 = (uint32_t)__builtin_amdgcn_readfirstlane((s_n - c) << 2);
            uint32_t at_off;
            asm("v_add_u32 %0, %1, %2" : "=v"(at_off) : "s"(top4), "v"(lane4));
            const char *const at = (const char *)s_list + at_off;
            uint32_t h[BS_GRP];
            /* (every ring slot is readable: the lanes past the end of a short, final trip are
             * cleared afterwards instead of being masked out of the loads) */
#pragma unroll
            for (int g = 0; g < BS_GRP; g++) h[g] = *(const uint32_t *)(at + g * BS_CAP * 4);
            const uint32_t meta = *(const uint32_t *)(at + BS_GRP * BS_CAP * 4);
            if (PARTIAL) {
#pragma unroll
                for (int g = 0; g < BS_GRP; g++) h[g] = (lane < c) ? h[g] : 0u;
            }
            uint32_t first = ffbl_or_ones(h[0]), total = 0u;
#pragma unroll
            for (int g = 1; g < BS_GRP; g++) first = min(first, ffbl_or_ones(h[g]) | (uint32_t)(g << 5));
#pragma unroll
            for (int g = 0; g < BS_GRP; g++) total = popc_add(h[g], total);
            uint32_t sel = first >> 5;
            const uint32_t bit = first & 31u;
#if defined(GKM_PROBE_VALU_F) || defined(GKM_PROBE_VALU_H) || defined(GKM_PROBE_LDS) || defined(GKM_PROBE_LDS64) || defined(GKM_PROBE_LAT)
            /* SENSITIVITY PROBES (experiments only, results unchanged): what one more full-rate / half-rate VALU
             * instruction, one more LDS operation, one more dependent LDS round trip per trip costs */
            {
                uint32_t dummy = total;
#ifdef GKM_PROBE_VALU_F
#pragma unroll
                for (int z = 0; z < GKM_PROBE_VALU_F; z++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(dummy) : "v"(first));
#endif
#ifdef GKM_PROBE_VALU_H
#pragma unroll
                for (int z = 0; z < GKM_PROBE_VALU_H; z++) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(dummy) : "v"(first));
#endif
#ifdef GKM_PROBE_LDS
#pragma unroll
                for (int z = 0; z < GKM_PROBE_LDS; z++) asm volatile("ds_add_u32 %0, %1" : : "v"(lane4), "v"(0u) : "memory");
#endif
#ifdef GKM_PROBE_LDS64 /* the same with 8 bytes per lane: is an LDS operation's cost its instruction or its bytes? */
                {
                    const unsigned long long z64 = 0ull;
                    const uint32_t lane8 = lane4 + lane4;
#pragma unroll
                    for (int z = 0; z < GKM_PROBE_LDS64; z++) asm volatile("ds_add_u64 %0, %1" : : "v"(lane8), "v"(z64) : "memory");
                }
#endif
#ifdef GKM_PROBE_LAT
#pragma unroll
                for (int z = 0; z < GKM_PROBE_LAT; z++) sel = (uint32_t)__builtin_amdgcn_ds_bpermute((int)lane4, (int)sel);
#endif
                asm volatile("" : : "v"(dummy));
            }
#endif
            const uint32_t ms = meta + sel; /* the word index w0 + sel <= W - 1 stays inside its 4 bits */
            uint32_t pslot4 = 0u, pc0b = 0u;
            if (BPERM) { /* every lane takes part (ds_bpermute_b32 reads 0 from lanes that EXEC masks out) */
                const int from = (int)((ms >> (META_LANE_SHIFT - 2)) & 0xFCu); /* source lane * 4 */
                /* ONE permute of (slot * 4 | c0b << 16) and two full-rate VALU operations to take it apart, not two
                 * permutes: the LDS pipe is busy two thirds of the time on gkmQC's shape (SQ_LDS_IDX_ACTIVE per CU against
                 * the kernel's cycles, profiles/r4_pmc_peaks.json): 392.6 -> 388.6 ms (profiles/r4_kernel_ab_trip_priority.txt) */
                const uint32_t both = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)my_both);
                pslot4 = both & 0xFFFFu;
                pc0b = both >> 16;
            }
            /* every record of a full trip holds a hit (only records with one are pushed or pushed again): no test */
            uint32_t cont = 0u;
            {
                /* is the same bit of the record's NEXT word set (the record still sits in the list where it was read)?
                 * Not for the group's last word (sel = 4: what follows is the origin word). */
                /* (PARTIAL: the lanes past the last record hold no hit word, `first` is all ones: keep their read inside the list) */
                const uint32_t nxt = *(const uint32_t *)(at + (((PARTIAL ? sel & 3u : sel) + 1u) << 9));
                static_assert(BS_CAP * 4 == 512, "word g of a record is g * 512 bytes on");
                cont = ((nxt >> bit) & 1u) & ~(first >> 7);
            }
            uint32_t took = 0u; /* 1: the visit resolved the hit's neighbour on the diagonal too */
            if (!PARTIAL || total) took = resolve(ms, bit, pslot4, pc0b, cont);
            total -= took;
            s_n -= c;
            const unsigned long long more = __ballot(total > 1u);
            if (more) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(more >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)more, 0u));
                if (total > 1u) {
                    char *const to = (char *)s_list + ((rank + (uint32_t)s_n) << 2);
#pragma unroll
                    for (int g = 0; g < BS_GRP; g++) *(uint32_t *)(to + g * BS_CAP * 4) = h[g];
                    *(uint32_t *)(to + BS_GRP * BS_CAP * 4) = meta;
                    atomicXor((uint32_t *)(to + sel * (uint32_t)(BS_CAP * 4)), 1u << bit); /* ds_xor_b32: that hit is done */
                    if (took) atomicXor((uint32_t *)(to + (sel + 1u) * (uint32_t)(BS_CAP * 4)), 1u << bit);
                }
                s_n += (int)__popcll(more);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        /* GROUP: one trip over the `c` two-word records on top of the list.  A record says: some of the five windows
         * (bit row b, words w0 .. w0+4) of source lane r are hits.  Lane positions i0 .. i0+4 (i0 = 10 b + w0) are five
         * CONSECUTIVE l-mers of the row against five consecutive l-mers q .. q+4 of the column strand, and the two 16-base
         * windows that the hit path fetches anyway -- one v_alignbit_b32 per side -- hold all of them (5 + L - 1 <= 16
         * bases): the mismatch count of window k is the popcount of a bit field of ONE folded XOR word.  So the visit
         * resolves every hit of the (bit row, group) at once: no search through five hit words, no copy of the record
         * back with one bit cleared per hit, and the counting loop pushes 8 bytes per record instead of 24 (round 5: the
         * pushes, two per shift whatever the hits, cost 14 % of config 2's kernel and 23 % of gkmQC's shape when doubled
         * -- profiles/r5_trip_sensitivity.txt).
         * What makes every window of a pushed group safe to evaluate without its hit bit:
         *   row side     the lanes of a same-length problem own window counts that are multiples of 5 (gkm_pack.h
         *                own_mult), so a group is owned whole or not at all; windows past the row's last l-mer (the
         *                row's last lane) read the zero bytes behind the positional weight table;
         *   column side  the packed strands are CYCLIC (k_pack_strands), so a window that runs over the strand's end is
         *                the very l-mer the cyclic bit planes of the counting loop compared; it is not an l-mer of the
         *                sequence and reads a zero weight (the L - 1 zero bytes); a window PAST the end (q + k >= T) is
         *                the strand's k-th l-mer again and reads its weight from the five bytes behind / before the
         *                zeros (k_build_postab);
         *   m <= d       is tested per window (EXEC-masked ds_add): unlike the single-hit path this one looks at
         *                windows the counting loop did not flag. */
        auto trip_group = [&](auto partial_tag, int c) {
            constexpr bool PARTIAL = decltype(partial_tag)::value;
            __builtin_amdgcn_s_setprio(GKM_TRIP_PRIO);
            const uint32_t top4 = (uint32_t)__builtin_amdgcn_readfirstlane((s_n - c) << 2);
            uint32_t at_off;
            asm("v_add_u32 %0, %1, %2" : "=v"(at_off) : "s"(top4), "v"(lane4));
            const char *const at = (const char *)s_list + at_off;
            uint32_t any = *(const uint32_t *)at;
            const uint32_t ms = *(const uint32_t *)(at + BS_CAP * 4);
            if (PARTIAL) any = (lane < c) ? any : 0u;
            const uint32_t bit = ffbl_or_ones(any);
            const uint32_t rest = any & (any - 1u); /* the other bit rows of the group with a hit: back to the list */
            const uint32_t lane128 = ms & (63u << META_LANE_SHIFT);
            /* PGROUP: the source lane's mask of piece-start bit rows, from that lane's register.  EVERY lane takes part
             * (ds_bpermute_b32 reads 0 from lanes that EXEC masks out, and in a partial trip the source lane of a live
             * record may well be a lane without a record) */
            uint32_t lm = 0u;
            if (PGROUP) lm = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lane128 >> 5), (int)my_lmask);
            if (!PARTIAL || any) {
                const uint32_t i0 = __umul24(bit, (uint32_t)W) + (ms & 15u);
                uint32_t slot4, ia, nv = 5u; /* row slot * 4; LDS address of the row l-mer's weight; owned windows from i0 on */
                if (PGROUP) {
                    /* the piece of the source lane that owns bit row `bit`; its entry = slot | centre offset << 7 | owned
                     * windows << 20 */
                    const uint32_t below = lm & (0xFFFFFFFFu >> (31u - bit)); /* piece starts at or below the bit row */
                    const uint32_t k = (uint32_t)__builtin_popcount(below) - 1u;
                    const uint32_t b0 = 31u - (uint32_t)__builtin_clz(below);
                    const uint32_t lp = *(const uint32_t *)((const char *)lpiece + ((lane128 >> 3) + (k << 2)));
                    slot4 = (lp << 2) & 0x1FCu;
                    const uint32_t c0b = (lp >> 7) & 0x1FFFu;
                    /* a group is owned whole or not at all (piece counts are multiples of five) EXCEPT in a row's last
                     * piece, whose windows end where the row does: the bytes of the windows behind are masked below */
                    nv = (lp >> 20) - (i0 - __umul24(b0, (uint32_t)W));
                    ia = (i0 - c0b) + s_rowbase;
                } else {
                    slot4 = (ms >> (META_SLOT_SHIFT - 2)) & 0xFCu;
                    ia = __umul24((ms >> META_PIECE_SHIFT) & 7u, (uint32_t)A.cap) + i0 + s_rowbase;
                }
                const uint32_t x = i0 + (ms >> 21);
                uint32_t q;
                if ((uint32_t)T >= (uint32_t)(32 * W)) q = min(x - (uint32_t)T, x); /* x < 2T (wave-uniform test) */
                else q = mod_small(x, (uint32_t)T, rcpT);
                const uint32_t *rw = (const uint32_t *)(rowpk_tile + lop3<0xEA>(i0 >> 2, ~3u, lane128));
                typedef const uint32_t __attribute__((address_space(3))) *lds_words;
                const uint32_t cwo = lop3<0xEA>(q >> 1, ~7u, s_strand4);
                const lds_words cw = COL_BASE_FOLDS ? (lds_words)(uintptr_t)cwo : (lds_words)(uintptr_t)((uint32_t)(STATIC_WORDS * 4) + cwo);
                /* the five weight bytes of either side as (four bytes, one byte): an aligned pair of LDS words around the
                 * first byte, funnel-shifted to it.  Row: bytes ia .. ia+4 of the positional table (l-mers p .. p+4).
                 * Column: ib .. ib+4 on the forward strand; the reverse strand's weights are the forward ones mirrored
                 * (libgkm.c:924), bytes ib, ib-1, .., ib-4: fetched from ib-4 up and turned round by v_perm_b32 with
                 * strand-uniform selectors. */
                const uint32_t ib = (q ^ s_wsign) + s_wbase;      /* this window's byte */
                const uint32_t il = ib - s_wback;                 /* the lowest of the five addresses (s_wback = 4 on the reverse strand) */
                /* (ia, ib are LDS ADDRESSES here: s_rowbase and s_wbase include the dynamic LDS's start, below) */
                const lds_words pa = (lds_words)(uintptr_t)(ia & ~3u);
                const lds_words pb = (lds_words)(uintptr_t)(il & ~3u);
                const uint32_t a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
                const uint32_t sha = (ia & 3u) << 3, shb = (il & 3u) << 3;
                uint32_t wa4 = __builtin_amdgcn_alignbit(a1, a0, sha), wa1 = a1 >> sha;
                if (PGROUP) { /* windows nv .. 4 of the group lie behind the row's last l-mer: weight 0 */
                    wa4 &= 0xFFFFFFFFu >> (32u - 8u * min(nv, 4u));
                    wa1 = nv >= 5u ? wa1 : 0u;
                }
                const uint32_t lo = __builtin_amdgcn_alignbit(b1, b0, shb), hi = b1 >> shb;
                const uint32_t wb4 = __builtin_amdgcn_perm(hi, lo, s_perm4), wb1 = __builtin_amdgcn_perm(hi, lo, s_perm1);
                /* (v_alignbit_b32 uses the low 5 bits of its count: 2 i0 mod 32 = 2 (i0 mod 16)) */
                const uint32_t ea = __builtin_amdgcn_alignbit(rw[1], rw[0], twice(i0));
                const uint32_t eb = __builtin_amdgcn_alignbit(cw[2], cw[0], twice(q));
                uint32_t t = ea ^ eb;
                t = (t | (t >> 1)) & 0x55555555u; /* one bit per mismatching base of the 16 */
                static_assert(5 + L - 1 <= 16, "the five windows of a group lie inside one 16-base window");
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const uint32_t m = (uint32_t)__builtin_popcount(__builtin_amdgcn_ubfe(t, 2u * k, 2u * L));
                    const uint32_t wa = k < 4 ? ((wa4 >> (8 * k)) & 0xFFu) : (wa1 & 0xFFu);
                    const uint32_t wb = k < 4 ? ((wb4 >> (8 * k)) & 0xFFu) : (wb1 & 0xFFu);
                    if (m <= (uint32_t)D) /* LDS atomic: ds_add_u32 (a window that is no l-mer adds 0) */
                        atomicAdd((uint32_t *)((char *)accl + (m * (uint32_t)(NSLOT * 4) + slot4)), wa * wb);
                }
            }
            s_n -= c;
            const unsigned long long more = __ballot(rest != 0u);
            if (more) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(more >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)more, 0u));
                if (rest != 0u) {
                    char *const to = (char *)s_list + ((rank + (uint32_t)s_n) << 2);
                    *(uint32_t *)to = rest;
                    *(uint32_t *)(to + BS_CAP * 4) = ms;
                }
                s_n += (int)__popcll(more);
            }
            __builtin_amdgcn_s_setprio(0);
        };

        /* Resolve the hit list in FULL trips of 64 records with every lane busy: each record gives up
         * its first hit (lowest bit of its first non-empty word), a record with more hits is appended
         * again.  Fewer than one trip's worth of records waits in the list; the last call of a column
         * (final) empties it.
         * No select chains: the position of the first hit is min over the words of ffbl(word) | 32 g
         * (v_ffbl_b32 gives all ones for an empty word, so empty words lose the min), the number of
         * hits left is a popcount sum, and a record that goes back to the list is copied unchanged and
         * then loses that hit by ONE LDS xor on the copy (the LDS operations of a wave execute in order). */
        auto one_trip = [&](auto partial_tag, int c) {
            if constexpr (GROUP) trip_group(partial_tag, c);
            else trip(partial_tag, c);
        };
        auto trips = [&](bool final) {
            while (s_n >= BS_TRIP) one_trip(std::false_type(), BS_TRIP);
            if (final)
                while (s_n > 0) {
                    if (s_n >= BS_TRIP) one_trip(std::false_type(), BS_TRIP);
                    else one_trip(std::true_type(), s_n);
                }
        };

        for (int strand = 0; strand < 2; strand++) {
            s_strand4 = (uint32_t)strand * 4u + (COL_BASE_FOLDS ? (uint32_t)(STATIC_WORDS * 4) : 0u);
            s_wsign = strand ? ~0u : 0u;
            s_wstep = strand ? ~0u : 1u; /* POSTAB: the next window's weight byte is the next / the previous one */
            /* GROUP: the five column weights start 4 bytes lower on the reverse strand and are turned round; selectors of
             * v_perm_b32(hi, lo): byte k of the result is byte sel_k of (hi:lo) */
            s_wback = strand ? 4u : 0u;
            s_perm4 = strand ? 0x01020304u : 0x03020100u;
            s_perm1 = strand ? 0x0c0c0c00u : 0x0c0c0c04u;
            if (POSTAB || PGROUP) s_wbase = (uint32_t)pkw * 8u + POSTAB_PAD + (uint32_t)(L - 1) + (strand ? (uint32_t)nB : 0u) + /* ~q = -q - 1 */
                                            (GROUP ? (uint32_t)(STATIC_WORDS * 4) : 0u);
            else s_wbase = (strand && !(nB & 1)) ? 1u : 0u;
            /* read-only, wave-uniform: address space 4 makes hipcc fetch these words with
             * scalar loads (s_load_dwordx*) into SGPRs instead of per-lane vector loads */
            const sgpr_words sbh = (sgpr_words)(A.sb + ((size_t)(j * 2 + strand) * 2) * A.xw);
            const sgpr_words sbl = sbh + A.xw;
            for (int d0 = 0; d0 < T; d0 += BS_DU) {
                /* (copying the words to VGPRs once instead of using them as SGPR operands was measured
                 * slower: 119-129 ms against 111.6 ms on config 2; requesting the next block's words one
                 * block ahead changes nothing: 92.3 against 92.4 ms; round 5: touching the cache line 48 words ahead
                 * with a throw-away scalar load changes nothing either, not even in the small launches of an 8-way
                 * split where 1 wave in 10-20 is the first to read its column -- profiles/r5_small_launch_probe.txt) */
                /* the strand's window-validity plane (third SB plane) is not streamed: wrapped
                 * windows are rejected when a hit is resolved (gkm_bitslice.h window_hits) */
                uint32_t bh[BS_DU + W - 1], bl[BS_DU + W - 1];
#pragma unroll
                for (int i = 0; i < BS_DU + W - 1; i++) {
                    bh[i] = sbh[d0 + i];
                    bl[i] = sbl[d0 + i];
                }
#pragma unroll
                for (int u = 0; u < BS_DU; u++) {
                    if (d0 + u < T) {
                        uint32_t hit[W];
                        window_hits<W, L, D>(Ahi, Alo, AV, bh + u, bl + u, (const uint32_t *)nullptr, hit);
                        const uint32_t vbase = lane_tag | pack_meta(d0 + u, 0, 0);
#pragma unroll
                        for (int w0 = 0; w0 < W; w0 += BS_GRP) {
                            /* wave-level compaction at the source, once per group of BS_GRP words: the
                             * lanes with a hit in the group append (words, origin) to the list at tail
                             * + their rank among the hit lanes (ballot + mbcnt); EXEC-masked stores, no
                             * divergent control flow */
                            uint32_t any = hit[w0];
#pragma unroll
                            for (int g = 1; g + 1 < BS_GRP; g += 2) any = lop3<TT_OR3>(any, hit[w0 + g], hit[w0 + g + 1]);
                            if (BS_GRP % 2 == 0) any |= hit[w0 + BS_GRP - 1];
                            const unsigned long long mask = __ballot(any != 0u);
                            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                            if (any != 0u) {
                                char *const at = (char *)s_list + (((uint32_t)rank + (uint32_t)s_n) << 2);
                                if (GROUP) { /* 8 bytes per record instead of 24: one ds_write2st64_b32 */
                                    *(uint32_t *)at = any;
                                } else {
#pragma unroll
                                    for (int g = 0; g < BS_GRP; g++) *(uint32_t *)(at + g * BS_CAP * 4) = hit[w0 + g];
                                }
                                *(uint32_t *)(at + (LIST_ARRAYS - 1) * BS_CAP * 4) = vbase | (uint32_t)w0;
                            }
                            s_n += (int)__popcll(mask);
                            if (s_n >= BS_TRIP) trips(false);
                        }
                    }
                }
            }
            trips(true); /* the list is empty before the other strand starts: see s_strand4 */
        }

        /* epilogue: one lane per row slot of the tile */
#pragma unroll
        for (int k = 0; k < NE; k++) {
            /* (row and output row of the slot are read here, not kept in registers through the sweep) */
            const int rs = k * 64 + lane;
            const int row = rs < nrows ? A.tile_row[tile * gkmpack::MAX_ROWS + rs] : -1;
            if (row < 0 || (j > row && !A.out.write_all)) continue;
            /* the profile: both copies where there are two (uint32 addition: the int32 wrap-around of the
             * reference's accumulator, libgkm.c:338, is kept) */
            uint32_t prof[D + 1];
#pragma unroll
            for (int m = 0; m <= D; m++) prof[m] = accl[m * NSLOT + rs] + (two_copies ? accl[m * NSLOT + rs + NSLOT / 2] : 0u);
            /* sum_m c_m P_m in ascending m from 0.0 (libgkm.c:576-582) */
            double g = 0.0;
#pragma unroll
            for (int m = 0; m <= D; m++) g += A.c[m] * (double)(int32_t)prof[m];
            const int64_t r = A.out.local_rows ? A.tile_out[tile * gkmpack::MAX_ROWS + rs] : row;
            if (A.out.diag && j == row) A.out.diag[row] = g;
            if (A.S) A.S[(A.tile_soff[tile] + (j - A.tile_cbeg[tile])) * NSLOT + rs] = g;
            if (A.out.P) {
#pragma unroll
                for (int m = 0; m <= D; m++)
                    A.out.P[(r * A.out.ldp + j) * (D + 1) + m] = (int32_t)prof[m];
            }
        }
    }
}

/* ------------------------------------------------- the instantiation table */
template <int W, int PACKED>
static bs_kernel_t pick_bitslice(int L, int d)
{
#define GKM_BS(LL, DD) \
    if (L == LL && d == DD) return k_gram_bitslice<W, LL, DD, PACKED>;
    /* every (L, d) with 3 <= L <= 12, d <= min(4, L - 1) (what bin/gkmqc.py:185 can ask for), plus the d > 4 pairs
     * where this kernel beats k_gram_direct -- see auto_takes_bitslice() below for where that is. */
#define GKM_BS_L(LL) GKM_BS(LL, 0) GKM_BS(LL, 1) GKM_BS(LL, 2) GKM_BS(LL, 3) GKM_BS(LL, 4)
    GKM_BS(3, 0) GKM_BS(3, 1) GKM_BS(3, 2)
    GKM_BS(4, 0) GKM_BS(4, 1) GKM_BS(4, 2) GKM_BS(4, 3)
    GKM_BS_L(5) GKM_BS_L(6) GKM_BS_L(7) GKM_BS_L(8) GKM_BS_L(9) GKM_BS_L(10) GKM_BS_L(11) GKM_BS_L(12)
    GKM_BS(11, 5) GKM_BS(12, 5) GKM_BS(12, 6)
#undef GKM_BS_L
#undef GKM_BS
    return nullptr;
}

bs_kernel_t gkm_pick_bitslice(int pk, int L, int d)
{
    switch (pk) {
    case 0: return pick_bitslice<10, 0>(L, d);
    case 1: return pick_bitslice<10, 1>(L, d);
    case 2: return pick_bitslice<10, 2>(L, d);
    case 3: return pick_bitslice<10, 3>(L, d);
    case 4: return pick_bitslice<10, 4>(L, d);
    }
    return nullptr;
}

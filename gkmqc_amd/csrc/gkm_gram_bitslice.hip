/*
 * gkm_gram_bitslice.hip -- the HOT kernel of the gkm kernel-matrix path on MI355X (gfx950): bit-sliced diagonal mismatch
 * profiles -> raw Gram values, tile-transposed (replaces the k-mer tree DFS of src/libgkm.c:315-387
 * and the per-row reduction of :553-589).  Written for CDNA4 only: 64-wide wavefronts, one wavefront per workgroup, the
 * column tables streamed through the scalar unit (SGPRs), a per-wave hit list in LDS.  The launch geometry (row packing,
 * work-item order, tables) is gkm_gram.hip's; the per-lane arithmetic is gkm_bitslice.h's (unit-tested on the CPU).
 */
#include "gkm_gram_bitslice.h"

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
#define GKM_BS_PACKED_WAVES 6 /* the several-pieces variants (ragged lengths): compiled for 6 they come out at 72 VGPRs and run 7 */
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
    /* PK = 4: problems whose sequences all have the same length (gkmQC's own 600-bp subsets, BASELINE configs 1-3): a row
     *         takes k = ceil(windows / 310) whole lanes, piece pi of it starts at sequence position pi * capacity, and all
     *         a trip needs of the source lane -- its row slot and pi -- rides in the record's origin word (9 spare bits,
     *         set once per wave): no piece table, no permute, and the row l-mer's weight comes from the column's own table
     *         by position (same length, same weights);
     *      1: everything else -- several pieces per lane (gkm_pack.h), up to 64 rows per tile; 2: up to 128 rows per tile.
     * (Rounds 2-5 also had one-piece variants for ragged lengths, PK = 0 / 3: piece entries in an LDS table / fetched by
     * ds_bpermute_b32, hits resolved one by one.  Once the group records below served ragged data through PK = 1 they were
     * left with L < 5 only, where the general kernel k_gram_direct now serves: git history has them.) */
    constexpr bool PACKED = PK == 1 || PK == 2;
    constexpr bool UNIF = PK == 4;
    static_assert(PACKED || UNIF, "PK = 4, 1 or 2");
    static_assert(L >= 5, "the L - 1 zero bytes either side of a weight table cover a group of five windows");
    using namespace gkmbs;
    /* LDS per wave.  STATIC, one array carved by hand:
     *   accl   [(D + 1) * NSLOT]     mismatch profiles [m][row slot]                              1-2.5 KB
     *   s_list [2][CAP]              the hit list: two-word group records (below)                 1 KB
     *   lpiece [64 * NP | 0]         piece entries of the several-pieces variants                 0-1 KB
     * DYNAMIC: the column's two 2-bit packed strands, interleaved word by word (2 * pkw words: 0.2 KB at 300 bp, 0.3 KB
     * at 600 bp), then the weight tables: the column's weights by l-mer position (gkm_gram_bitslice.h POSTAB_PAD; ~T + L
     * + 16 bytes), behind it the row side's -- none in the same-length variant (the rows read the column's table), the
     * centred distance table in the several-pieces variants.  What a visit
     * reads: two words of the column strand and the weight bytes from LDS; two words of the row lane's packed positions
     * (8 KB per tile -- 128 bytes per lane, of which 84 are used -- L1 resident: the waves of a CU work on the same
     * tile) from global memory. */
    extern __shared__ uint32_t s_dyn[];
    /* PACKED: lanes may hold several pieces (gkm_pack.h).
     * The several-pieces variant exists for 64 and for 128 row slots per tile: the profiles of 128 slots
     * (2.5 KB at d = 4) cost a wave per SIMD, so the host packs at most 64 rows into a tile unless
     * that would leave lanes empty (many rows shorter than half a lane). */
    constexpr int NP = PACKED ? gkmpack::MAX_PIECES : 1;   /* pieces per lane */
    constexpr int NSLOT = PK == 2 ? gkmpack::MAX_ROWS : 64; /* row slots per tile */
    /* GROUP RECORDS: a record of the hit list is TWO words -- the OR of the five hit words of a group of words (which bit
     * rows of the lane hold a hit somewhere in the group) and the origin -- and a trip finds the hits among the five
     * windows of (bit row, group) itself, from the packed strands it reads anyway (see `trip`). */
    constexpr int LIST_ARRAYS = 2;
    constexpr int ACC_WORDS = (D + 1) * NSLOT, LIST_WORDS = LIST_ARRAYS * BS_CAP;
    /* per piece of the several-pieces variants: row slot | centre offset << 7 | owned windows << 20 */
    constexpr int LPIECE_WORDS = PACKED ? 64 * NP : 0;
    constexpr int STATIC_WORDS = ACC_WORDS + LIST_WORDS + LPIECE_WORDS;
    __shared__ uint32_t s_mem[STATIC_WORDS];
    uint32_t *const accl = s_mem; /* mismatch profiles [m][row slot] */
    /* The hit list.  Word k of record i sits at s_list[k * BS_CAP + i]: k = 0 the OR of the lane's BS_GRP hit words for
     * BS_GRP consecutive words of a shift, k = 1 their origin (first word of the group, shift, row lane); the arrays are
     * a multiple of 64 dwords apart, so that the two stores of a push merge into one ds_write2st64_b32.  Compacting once
     * per group instead of once per word takes 3 VALU instructions per word out of the hot loop (config 2: 111.0 -> 96.2
     * ms in round 1). */
    uint32_t *const s_list = s_mem + ACC_WORDS;
    uint32_t *const lpiece = s_list + LIST_WORDS;
    static_assert((ACC_WORDS * 4) % 256 == 0, "the list's arrays stay 64-dword aligned (ds_write2st64_b32)");
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
    /* (several pieces: the piece-start mask stays in a register of its lane and a trip fetches it by ds_bpermute_b32 -- the
     * 256 bytes keep the wave inside four LDS allocation granules at 600 bp) */
    uint32_t my_lmask = 0u;
    if (PACKED) {
        my_lmask = A.lane_mask[tile * 64 + lane];
#pragma unroll
        for (int k = 0; k < NP; k++) lpiece[lane * NP + k] = A.lane_piece[(size_t)(tile * 64 + lane) * NP + k];
    }
    /* (UNIF: word 0 of the lane's piece entry holds its row slot and piece index where the origin word wants them) */
    const uint32_t lane_tag = ((uint32_t)lane << META_LANE_SHIFT) | (UNIF ? A.lane_piece[(size_t)(tile * 64 + lane) * 2] : 0u);
    const uint32_t lane4 = (uint32_t)lane << 2;
    const int pkw = A.pkw;
    /* dynamic LDS: the column's two packed strands first, interleaved word by word, the weight bytes behind them */
    uint32_t *const s_col = s_dyn;
    uint8_t *const s_wtab = (uint8_t *)(s_dyn + 2 * pkw);
    /* The COLUMN's weights sit in LDS by l-mer position with L - 1 zero bytes either side and five wrap bytes outside those
     * (gkm_gram_bitslice.h POSTAB_PAD), which rids a trip of every test but m <= d (below). */
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
         * nothing else; ones for the unweighted kernel types).  The COLUMN's weights BY POSITION (k_build_postab,
         * gkm_context.hip: built once per sequence, the wave copies dwords as it copies the strands): byte POSTAB_PAD + L - 1
         * + p = wt[p] = wd[|nB/2 - p|] for the l-mers p < nB, L - 1 zero bytes either side, five wrap bytes outside those.
         * A forward-strand window q reads byte .. + q; the reverse strand's weights are the forward ones mirrored,
         * wt_rc[q] = wt[nB-1-q] (libgkm.c:924): byte .. + nB - 1 - q.  The windows that wrap around the end of the strand
         * (q = nB .. T - 1: not l-mers, gkm_bitslice.h window_hits) land in the zero bytes: they add 0 and need no test.
         * (A distance-indexed table cannot do that: |nB/2 - q| of q = nB equals that of q = 0 when nB is even.)
         * The ROW side: the same-length variant reads the very same table by position (every sequence has the column's
         * length); the several-pieces variants read a CENTRED distance table behind it (A.ptw words on). */
        uint32_t s_rowbase; /* LDS address of the row side's weight byte for lane position 0 (same length) / distance 0 */
        for (int x = lane; x < A.ptw; x += 64) ((uint32_t *)s_wtab)[x] = A.postab[(size_t)j * A.ptw_stride + x];
        if (PACKED) {
            /* byte A.wdc_centre + s = wd[|s|] for the signed distance s of an l-mer to its sequence's centre l-mer, so that
             * five consecutive l-mers read five consecutive bytes (the host builds it: gkm_context.hip; it serves every row
             * length) */
            for (int x = lane; x < A.wdc_words; x += 64) ((uint32_t *)(s_wtab + A.ptw * 4))[x] = A.wdc[x];
            /* (the l-mer at lane position i0 has the signed distance i0 + 2048 - c0b: its byte is at i0 - c0b + this) */
            s_rowbase = (uint32_t)(STATIC_WORDS * 4) + (uint32_t)pkw * 8u + (uint32_t)A.ptw * 4u + (uint32_t)A.wdc_centre + 2048u;
        } else {
            s_rowbase = (uint32_t)(STATIC_WORDS * 4) + (uint32_t)pkw * 8u + POSTAB_PAD + (uint32_t)(L - 1);
        }
        /* strand-uniform scalars of the hit path (set at the top of each strand's sweep; the list is emptied between
         * the strands, so a trip only ever holds records of ONE strand and the strand costs it no instruction):
         *   s_strand4   byte offset of the strand's words in the interleaved column image (0 / 4)
         *   s_wsign 0 / ~0, s_wbase: the column weight's LDS byte is (q ^ s_wsign) + s_wbase = L-1 + q or L-1 + nB-1 - q
         *   behind the table's start; s_wback, s_perm4, s_perm1: the reverse strand's five bytes are fetched from 4 lower
         *   and turned round */
        uint32_t s_strand4 = 0u, s_wsign = 0u, s_wbase = 0u, s_wback = 0u, s_perm4 = 0x03020100u, s_perm1 = 0x0c0c0c04u;
#pragma unroll
        for (int m = 0; m <= D; m++)
            for (int rs = lane; rs < (two_copies ? NSLOT : nrows); rs += 64) accl[m * NSLOT + rs] = 0u;
        int s_n = 0; /* records in the hit list (wave-uniform) */

        /* A wave inside a trip issues AHEAD of the waves that are in the counting loop (s_setprio; back to 0 at the end of
         * the trip).  A trip is a chain of short instruction runs between LDS and memory round trips (record -> piece
         * entry -> row words -> column words and weights -> accumulate); at equal priority each run waits its turn behind
         * six waves of straight-line counting code, and the chain -- with the LDS list and the other lanes' hits waiting on
         * it -- stretches.  Round 4, same-run A/B (profiles/r4_kernel_ab_trip_priority.txt): config 2 75.2 -> 72.8 ms,
         * gkmQC's own shape 433.3 -> 396.0 ms, config 5 167.4 -> 152.7 ms; priority 1 and 3 do the same.  The total VALU
         * work is unchanged: this is issue ORDER, not instruction count.
         * Written for the ISSUE COST: on gfx950 only the plain two-operand integer operations and v_bitop3_b32 issue at the
         * full rate; v_bfe, v_mad_u32_u24, v_min, v_alignbit, v_ffbl, v_bcnt, compares, SDWA and anything with an SGPR
         * operand take longer (tools/valu_ops.hip) -- though inside a trip the difference is 8 %, and one LDS operation
         * costs what three VALU instructions do (profiles/r5_trip_sensitivity.txt).  Hence the layout of the origin word
         * (gkm_bitslice.h pack_meta: fields that are masked in place or shifted out of the top), 128 bytes per lane of
         * packed positions, the column's strands interleaved word by word, (a & const) | b as one v_bitop3_b32, the strand
         * as wave-uniform scalars.  Same arithmetic as resolve_hit_packed (gkm_bitslice.h), which the CPU tests run
         * against the oracle. */
        /* One trip over the `c` two-word records on top of the list (PARTIAL: c < BS_TRIP, the last trip of a strand).  A
         * record says: some of the five windows (bit row b, words w0 .. w0+4) of source lane r are hits.  Lane positions i0 .. i0+4 (i0 = 10 b + w0) are five
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
         *   m <= d       is tested per window (EXEC-masked ds_add): the visit looks at windows the counting loop did not
         *                flag. */
        auto trip = [&](auto partial_tag, int c) {
            constexpr bool PARTIAL = decltype(partial_tag)::value;
            __builtin_amdgcn_s_setprio(GKM_TRIP_PRIO);
            /* the c records on top: lane * 4 + a scalar (kept apart from the lane term: hipcc would fuse the shift into a
             * half-rate v_lshl_add_u32 and split the reads around a negative offset) */
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
            /* several pieces: the source lane's mask of piece-start bit rows, from that lane's register.  EVERY lane takes part
             * (ds_bpermute_b32 reads 0 from lanes that EXEC masks out, and in a partial trip the source lane of a live
             * record may well be a lane without a record) */
            uint32_t lm = 0u;
            if (PACKED) lm = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lane128 >> 5), (int)my_lmask);
            if (!PARTIAL || any) {
                const uint32_t i0 = __umul24(bit, (uint32_t)W) + (ms & 15u);
                uint32_t slot4, ia, nv = 5u; /* row slot * 4; LDS address of the row l-mer's weight; owned windows from i0 on */
                if (PACKED) {
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
                if (PACKED) { /* windows nv .. 4 of the group lie behind the row's last l-mer: weight 0 */
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

        /* Resolve the hit list in FULL trips of 64 records with every lane busy: each record gives up its lowest bit row
         * (all five windows of the group in it), a record with more bit rows is appended again.  Fewer than one trip's
         * worth of records waits in the list; the last call of a strand (final) empties it. */
        auto trips = [&](bool final) {
            while (s_n >= BS_TRIP) trip(std::false_type(), BS_TRIP);
            if (final)
                while (s_n > 0) {
                    if (s_n >= BS_TRIP) trip(std::false_type(), BS_TRIP);
                    else trip(std::true_type(), s_n);
                }
        };

        for (int strand = 0; strand < 2; strand++) {
            s_strand4 = (uint32_t)strand * 4u + (COL_BASE_FOLDS ? (uint32_t)(STATIC_WORDS * 4) : 0u);
            s_wsign = strand ? ~0u : 0u;
            /* the five column weights start 4 bytes lower on the reverse strand and are turned round; selectors of
             * v_perm_b32(hi, lo): byte k of the result is byte sel_k of (hi:lo) */
            s_wback = strand ? 4u : 0u;
            s_perm4 = strand ? 0x01020304u : 0x03020100u;
            s_perm1 = strand ? 0x0c0c0c00u : 0x0c0c0c04u;
            s_wbase = (uint32_t)(STATIC_WORDS * 4) + (uint32_t)pkw * 8u + POSTAB_PAD + (uint32_t)(L - 1) + (strand ? (uint32_t)nB : 0u); /* ~q = -q - 1 */
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
                            /* wave-level compaction at the source, once per group of BS_GRP words: the lanes with a hit in the
                             * group append (OR of the hit words, origin) to the list at tail + their rank among the hit lanes
                             * (ballot + mbcnt); one EXEC-masked ds_write2st64_b32, no divergent control flow.  (Round 5: these
                             * pushes -- two per shift whatever the hits -- cost 14 % of config 2's kernel and 23 % of gkmQC's
                             * shape when doubled, profiles/r5_trip_sensitivity.txt: 8 bytes per record, not 24.) */
                            uint32_t any = hit[w0];
#pragma unroll
                            for (int g = 1; g + 1 < BS_GRP; g += 2) any = lop3<TT_OR3>(any, hit[w0 + g], hit[w0 + g + 1]);
                            if (BS_GRP % 2 == 0) any |= hit[w0 + BS_GRP - 1];
                            const unsigned long long mask = __ballot(any != 0u);
                            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                            if (any != 0u) {
                                char *const at = (char *)s_list + (((uint32_t)rank + (uint32_t)s_n) << 2);
                                *(uint32_t *)at = any;
                                *(uint32_t *)(at + BS_CAP * 4) = vbase | (uint32_t)w0;
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
    /* every (L, d) with 5 <= L <= 12, d <= 4 (bin/gkmqc.py:181-185 allows 3 <= L <= 12: L = 3 and 4 take k_gram_direct, the
     * group records need L >= 5), plus the d > 4 pairs where this kernel beats k_gram_direct -- see bitslice_serves()
     * in gkm_gram.hip for where that is. */
#define GKM_BS_L(LL) GKM_BS(LL, 0) GKM_BS(LL, 1) GKM_BS(LL, 2) GKM_BS(LL, 3) GKM_BS(LL, 4)
    GKM_BS_L(5) GKM_BS_L(6) GKM_BS_L(7) GKM_BS_L(8) GKM_BS_L(9) GKM_BS_L(10) GKM_BS_L(11) GKM_BS_L(12)
    GKM_BS(11, 5) GKM_BS(12, 5) GKM_BS(12, 6)
#undef GKM_BS_L
#undef GKM_BS
    return nullptr;
}

bs_kernel_t gkm_pick_bitslice(int pk, int L, int d)
{
    switch (pk) {
    case 1: return pick_bitslice<10, 1>(L, d);
    case 2: return pick_bitslice<10, 2>(L, d);
    case 4: return pick_bitslice<10, 4>(L, d);
    }
    return nullptr;
}

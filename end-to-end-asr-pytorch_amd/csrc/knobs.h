// Tuning / experiment knobs of libasrk: environment variables are read ONCE (first asrk_init() or first
// use, whichever comes first, under std::call_once) into this immutable table; no entry point calls
// getenv() on its own and nothing changes the table afterwards, so the library holds no mutable mode
// state.  Arithmetic modes that callers legitimately switch per call (bf16x6 operand
// splitting on / off) are call ARGUMENTS (`flags`), not knobs.  UNSET = the variable is absent.
#pragma once

struct AsrkKnobs {
    static constexpr int UNSET = -0x7fffffff;
    // gemm.hip
    int gemm_noskinny;    // ASRK_GEMM_NOSKINNY (present = 1): never take the skinny-M weight-streaming path
    int skinny_sk;        // ASRK_SKINNY_SK: force the skinny path's K split
    int gemm_dbg;         // ASRK_GEMM_DBG: phase-skip mask of gemm_f32_fast (timing experiments)
    int gemm_nofast;      // ASRK_GEMM_NOFAST (present = 1)
    // gemm_split.hip
    int split_w256;       // ASRK_SPLIT_W256: 0 = never the 128x256-tile kernel, 1 (default) when the launch has >= 2 tiles per CU, 2 = whenever N >= 512
    int split_tail;       // ASRK_SPLIT_TAIL: 0 = the 128x256 kernel also takes a mostly empty last round (no 128x128 tail launch)
    // lstm_rec.hip
    int fwd_mt, fwd_nt;   // ASRK_FWD_MT / ASRK_FWD_NT: force a forward tile
    int rec_bf_mt4;       // ASRK_REC_BF_MT4: 0 = no 16-unit x 16-row forward plan at H = 1024
    int bwd_ub, bwd_nt;   // ASRK_BWD_UB / _NT: force a BPTT tile
    int fwd_poll, fwd_presleep;   // ASRK_FWD_POLL, ASRK_FWD_PRESLEEP (x64 cycles; default 8 since the publish-first store order of round 6, 16 before)
    int bwd_poll, bwd_presleep;   // ASRK_BWD_POLL (default 1), ASRK_BWD_PRESLEEP
    int dbg_noload;       // ASRK_DBG_NOLOAD (present = 1)
    // every file with float atomics
    int deterministic;    // ASRK_DETERMINISTIC=1: bit-reproducible results run to run - no f32 atomics whose order can
                          // vary: GEMMs never split K across workgroups, column sums / LayerNorm parameter gradients use
                          // one row chunk, the cross-entropy sum and the embedding gradient run in a fixed order
                          // (the reference's CPU path is reproducible; the default trades that for speed)

    int get(int v, int dflt) const { return v == UNSET ? dflt : v; }
    bool is_set(int v) const { return v != UNSET; }
};

const AsrkKnobs &asrk_knobs_();

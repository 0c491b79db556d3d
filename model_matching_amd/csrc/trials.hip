// trials.hip -- N independent StoCS trials in ONE set of launches (stocs_run_trials).
//
// The reference runs one trial per process: 100 base attempts, their congruent sets, <= 200 candidates per base, the best
// candidate (reference src/stocs_match_one_object.cpp:81-165).  A trial of a 640x480 frame is a few hundred microseconds of
// device work spread over ~40 launches, on a chip that could run a hundred of them side by side -- and BASELINE config 4 asks
// for 64 of them.  Here the trials of a batch share every launch:
//   phase 1  class mode: n_trials x n_attempts workgroups of class_attempts_kernel; instance mode: a pair of workgroups per
//            trial on its own copy of the mutable image-space state (sample.hip, sample_trials);
//   phase 2  the trials' base sets are concatenated and go through stocs_find_congruent_all as ONE base set: every structure
//            there is keyed by base (gather segments, (base, cell) sort keys, run tables, per-base quad counts), so a base neither
//            knows nor cares whose trial it belongs to;
//   phase 3  stocs_make_transforms over the concatenation -- each base draws its <= max subset with the seed of ITS trial and
//            under its slot in that trial (transform.hip), the candidates come out trial by trial;
//   phase 4  one scoring launch over all candidates, then compute_best_transform (stocs.cpp:982-1004) per trial
//            (trial_best_kernel); instance mode scores trial by trial, each against its own decayed class probabilities (Q8).
// A batch that would not fit -- 32-bit (base, cell) keys, 64-bit packed quads, the memory ceiling -- is cut into pieces of
// consecutive trials, each piece one set of launches.  Every trial's bases, congruent sets, candidates and winner are bit for
// bit what stocs_reset_trial + stocs_sample_bases + stocs_find_congruent_all + stocs_make_transforms + stocs_verify_all give
// for the same seed (tests/test_trials_gpu.py).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "stocs_ctx.h"

namespace stocs {

struct TrialBatch {
    int mode = 0, nT = 0, nA = 0, max_per_base = 0;
    bool keep = false;
    std::vector<uint64_t> seeds;
    std::vector<BaseOut> res;                                   // nT * nA attempts
    std::vector<stocs_trial_result> out;                        // nT
    std::vector<std::vector<long long> > quads;                 // per trial: congruent sets of each valid base
    std::vector<std::vector<float> > T, P, lcp;                 // per trial, when details are kept
    std::vector<std::vector<int32_t> > cbase;
    int pieces = 0;
};

// compute_best_transform (stocs.cpp:982-1004) of every trial of a piece: strict > from 0 -- the first maximum wins, a trial
// whose scores are all 0 has no pose -- over the trial's own stretch of the score array, and the winner's camera-frame pose
// next to it.  One workgroup per trial; out18[t] = (key lo, key hi, pose[16]) with the candidate index counted inside the trial.
__global__ __launch_bounds__(256) void trial_best_kernel(const float* __restrict__ lcp, const float* __restrict__ P, const int32_t* __restrict__ cand_off,
                                                         float* __restrict__ out18) {
    __shared__ unsigned long long sh[4];
    const int t = blockIdx.x;
    const int i0 = cand_off[t], i1 = cand_off[t + 1];
    unsigned long long k = 0;
    for (int i = i0 + (int)threadIdx.x; i < i1; i += 256) {
        const float s = lcp[i];
        if (s > 0.0f) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)(i - i0));
            k = key > k ? key : k;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(k, off, 64); k = o > k ? o : k; }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = k;
    __syncthreads();
    for (int w = 0; w < 4; ++w) k = sh[w] > k ? sh[w] : k;
    float* o = out18 + (size_t)t * 18;
    if (threadIdx.x == 0) { o[0] = __uint_as_float((uint32_t)(k & 0xFFFFFFFFull)); o[1] = __uint_as_float((uint32_t)(k >> 32)); }
    if (threadIdx.x < 16) {
        const uint32_t id = 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull);
        o[2 + threadIdx.x] = k ? P[((size_t)i0 + id) * 16 + threadIdx.x] : 0.0f;
    }
}

// candidate -> trial of the piece (candidates come out trial by trial: a binary search in the per-trial offsets)
__global__ __launch_bounds__(256) void cand_trial_kernel(const int32_t* __restrict__ cand_off, int n_trials, int first_trial, int n, int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = n_trials - 1;           // last t with cand_off[t] <= i
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (cand_off[mid] <= i) lo = mid; else hi = mid - 1; }
    out[i] = first_trial + lo;
}

static double now_ms() { return CallTiming::now_s() * 1e3; }

}  // namespace stocs

using namespace stocs;

extern "C" {

void stocs_internal_free_trials(stocs_ctx* c) {
    if (c && c->trials) { delete (TrialBatch*)c->trials; c->trials = NULL; }
}

int stocs_run_trials(stocs_ctx* c, int mode, int n_trials, const uint64_t* seeds, int n_attempts, float dispersion, int max_per_base, int keep_details,
                     stocs_trial_result* out) {
    if (!c || n_trials < 0 || n_attempts < 0 || max_per_base <= 0 || (mode != 0 && mode != 1) || (n_trials && !seeds)) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    const double t_entry = now_ms();
    if (!c->index.built) { set_error("stocs_run_trials: PPF index not built"); return STOCS_ERR_STATE; }
    if (mode == 1 && n_attempts > 254) { set_error("instance mode labels segments with a u8 (<= 254 attempts, Q14)"); return STOCS_ERR_INVALID; }
    if ((long long)n_trials * (long long)std::max(n_attempts, 1) > (1ll << 24)) { set_error("stocs_run_trials: more than 2^24 attempts in one batch"); return STOCS_ERR_INVALID; }
    // every trial starts from the class probabilities given at construction (instance-mode sampling decays them in place) and
    // from an empty base set: what stocs_reset_trial does -- needed only when an earlier call left the prior decayed
    if (c->h_sprob != c->h_sprob0) { const int rc0 = stocs_reset_trial(c); if (rc0) return rc0; }
    clear_trial_batch(c);
    c->bases.clear(); c->quad_off.clear(); clear_candidates(c);
    c->best_lcp = 0; c->best_index = -1; c->last_segment.clear();
    if (!c->trials) c->trials = new TrialBatch();
    TrialBatch* B = (TrialBatch*)c->trials;
    const int nT = n_trials, nA = n_attempts;
    B->mode = mode; B->nT = nT; B->nA = nA; B->max_per_base = max_per_base; B->keep = keep_details != 0; B->pieces = 0;
    B->seeds.assign(seeds, seeds + nT);
    B->res.assign((size_t)nT * (size_t)nA, BaseOut());
    B->out.assign((size_t)nT, stocs_trial_result());
    B->quads.assign((size_t)nT, std::vector<long long>());
    B->T.assign((size_t)nT, std::vector<float>()); B->P = B->T; B->lcp = B->T;
    B->cbase.assign((size_t)nT, std::vector<int32_t>());
    for (int t = 0; t < nT; ++t) { memset(&B->out[(size_t)t], 0, sizeof(stocs_trial_result)); B->out[(size_t)t].best_index = -1; }
    CallTiming& TM = c->timing[3];
    TM.begin();
    double ms_cong = 0, ms_xf = 0, ms_ver = 0, ms_asm = 0, ms_res = 0;
    const double ms_setup = now_ms() - t_entry;
    if (nT == 0) return STOCS_OK;
    if (nA > 0 && c->nS > 0) {
        const int rc = sample_trials(c, mode, nT, seeds, nA, dispersion, B->res.data(), &c->snrmw_trial0, &c->snrmw_stride);
        if (rc) { clear_trial_batch(c); B->nT = 0; B->out.clear(); return rc; }
    }
    const float4* snrmw0 = c->snrmw_trial0;
    TM.lap("sampling: every attempt of every trial in one launch + read-back");
    // ---- how many bases one set of launches can take (congruent.hip: 32-bit (base, cell) sort keys with the run table, packed
    //      64-bit quads with the base above the four model ids) ----
    long long max_bases = 1 << 20;
    {
        const float eps_unit = c->prm.distance_threshold / c->ratio;
        const int gridDepth = (int)(-log2f(eps_unit));
        const int egSize = (int)pow(2.0, (double)gridDepth);
        const long long NC = (long long)egSize * egSize * egSize;
        int id_bits = 1, cell_bits = 1;
        while ((1 << id_bits) < c->nM) id_bits++;
        if (NC > 0 && NC < (1ll << 31)) {
            while (cell_bits < 40 && (((unsigned long long)1 << cell_bits) - 1ull) < (unsigned long long)NC) cell_bits++;
            if (cell_bits < 32) max_bases = std::min(max_bases, 1ll << (32 - cell_bits));
            max_bases = std::min(max_bases, std::max(1ll, (1ll << 27) / NC));
        }
        if (4 * id_bits < 64) max_bases = std::min(max_bases, 1ll << std::min(20, 64 - 4 * id_bits));
        max_bases = std::max(max_bases, (long long)std::max(nA, 1));     // a single trial always goes through (as it does alone)
    }
    size_t max_bytes = (size_t)48 << 30;      // of 288 GB: a Cm trial is ~0.9 GB of pair lists, and at Cm a piece is then what the 64-bit quads can key (40 trials)
    if (const char* e = getenv("STOCS_TRIALS_MAX_MB")) max_bytes = (size_t)std::max(1, atoi(e)) << 20;
    int piece_cap = nT;
    if (const char* e = getenv("STOCS_TRIALS_PER_PIECE")) piece_cap = std::max(1, atoi(e));   // (tests: forces several pieces)
    std::vector<int32_t> n_valid((size_t)nT, 0);
    for (int t = 0; t < nT; ++t) for (int a = 0; a < nA; ++a) n_valid[(size_t)t] += B->res[(size_t)t * nA + a].valid ? 1 : 0;
    // A HIP error inside a piece leaves through the common epilogue below (batch state cleared, context reset, the batch record
    // marked invalid) instead of returning from the middle of the loop with a half-filled record that the getters would serve.
#define TRIALS_HIP_TRY(expr) { const hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); rc = STOCS_ERR_HIP; break; } }
    int rc = STOCS_OK;
    for (int t0 = 0; t0 < nT && !rc;) {
        int t1 = t0;
        long long nb = 0;
        while (t1 < nT && t1 - t0 < piece_cap && (t1 == t0 || nb + n_valid[(size_t)t1] <= max_bases)) { nb += n_valid[(size_t)t1]; ++t1; }
        int64_t total_quads = 0;
        for (;;) {   // the piece t0 .. t1: halved until its pair lists fit the ceiling
            const double t_asm = now_ms();
            c->bases.clear(); c->base_seed.clear(); c->base_local.clear(); c->trial_first_base.clear(); c->trial_cand_off.clear();
            c->quad_off.clear(); clear_candidates(c);
            for (int t = t0; t < t1; ++t) {
                c->trial_first_base.push_back((int32_t)c->bases.size());
                int slot = 0;
                for (int a = 0; a < nA; ++a) {
                    const BaseOut& r = B->res[(size_t)t * nA + a];
                    if (!r.valid) continue;
                    BaseRec br;
                    for (int k = 0; k < 4; ++k) br.ids[k] = r.ids[k];
                    br.inv1 = r.inv[0]; br.inv2 = r.inv[1];
                    c->bases.push_back(br);
                    c->base_seed.push_back(seeds[t]);
                    c->base_local.push_back(slot++);
                }
            }
            c->trial_first_base.push_back((int32_t)c->bases.size());
            const double ta = now_ms();
            ms_asm += ta - t_asm;
            int too_big = 0;
            rc = stocs_internal_find_congruent(c, &total_quads, t1 - t0 > 1 ? max_bytes : 0, &too_big);   // (one trial alone always goes, as it does through the single calls)
            ms_cong += now_ms() - ta;
            if (rc || !too_big) break;
            if (t1 - t0 <= 1) { set_error("stocs_run_trials: the pair lists of one trial exceed 2^32 entries"); rc = STOCS_ERR_CAPACITY; break; }
            t1 = t0 + (t1 - t0) / 2;
            piece_cap = t1 - t0;          // the later pieces of this batch will be about as big
        }
        if (rc) break;
        B->pieces++;
        const int nTp = t1 - t0;
        double ta = now_ms();
        int n_cand = 0;
        if ((rc = stocs_make_transforms(c, max_per_base, 0, &n_cand))) break;
        ms_xf += now_ms() - ta;
        ta = now_ms();
        if ((int)c->trial_cand_off.size() != nTp + 1) c->trial_cand_off.assign((size_t)nTp + 1, 0);   // (a piece without any base)
        // ---- verification: every candidate of the piece scored, then the arg-max of each trial ----
        std::vector<float> out18((size_t)nTp * 18, 0.0f);
        if (n_cand > 0) {
            // scratch of this step (the transform jobs of the piece are done with the area): per-trial candidate offsets | per-trial
            // results | (instance mode) the trial of every candidate
            const bool per_trial_weights = mode == 1 && snrmw0 != NULL;
            const size_t ob = (((size_t)nTp + 1) * 4 + 255) & ~(size_t)255, rb = (((size_t)nTp * 18 * 4) + 255) & ~(size_t)255,
                         tb = per_trial_weights ? (((size_t)n_cand * 4 + 255) & ~(size_t)255) : 0;
            if ((rc = ensure_scratch(c, ob + rb + tb + 256))) break;
            if ((rc = ensure_pinned(c, (size_t)PIN_VAR + ob + rb + 256))) break;
            int32_t* d_off = (int32_t*)c->d_scratch;
            float* d_out = (float*)((char*)c->d_scratch + ob);
            int32_t* d_ct = (int32_t*)((char*)c->d_scratch + ob + rb);
            int32_t* off_pin = (int32_t*)((char*)c->h_pin + PIN_VAR);
            float* out_pin = (float*)((char*)c->h_pin + PIN_VAR + ob);
            memcpy(off_pin, c->trial_cand_off.data(), 4 * ((size_t)nTp + 1));
            TRIALS_HIP_TRY(hipMemcpyAsync(d_off, off_pin, 4 * ((size_t)nTp + 1), hipMemcpyHostToDevice, c->stream));
            if (!per_trial_weights) {
                if ((rc = launch_lcp(c, cand_T(c), n_cand, cand_lcp(c), NULL, NULL, NULL, 0))) break;
            } else {
                // instance mode: every candidate against the class probabilities as ITS trial's sampling decayed them (Q8) -- still ONE
                // launch: the kernel picks the trial's copy of the scene normals + weights per candidate (LcpArgs::cand_trial)
                hipLaunchKernelGGL(cand_trial_kernel, dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, c->stream, (const int32_t*)d_off, nTp, t0, n_cand, d_ct);
                TRIALS_HIP_TRY(hipGetLastError());
                c->snrmw_override = snrmw0; c->lcp_cand_trial = d_ct;
                rc = launch_lcp(c, cand_T(c), n_cand, cand_lcp(c), NULL, NULL, NULL, 0);
                c->snrmw_override = NULL; c->lcp_cand_trial = NULL;
                if (rc) break;
            }
            hipLaunchKernelGGL(trial_best_kernel, dim3((unsigned)nTp), dim3(256), 0, c->stream, (const float*)cand_lcp(c), (const float*)cand_P(c), (const int32_t*)d_off, d_out);
            TRIALS_HIP_TRY(hipGetLastError());
            TRIALS_HIP_TRY(hipMemcpyAsync(out_pin, d_out, (size_t)nTp * 18 * 4, hipMemcpyDeviceToHost, c->stream));
            TRIALS_HIP_TRY(hipStreamSynchronize(c->stream));
            memcpy(out18.data(), out_pin, (size_t)nTp * 18 * 4);
            c->cands_stale = true;
        }
        ms_ver += now_ms() - ta;
        ta = now_ms();
        // ---- results of the piece ----
        std::vector<float> hT, hP, hL; std::vector<int32_t> hB;
        if (B->keep && n_cand > 0) {
            hT.resize((size_t)n_cand * 16); hP.resize((size_t)n_cand * 16); hL.resize((size_t)n_cand); hB.resize((size_t)n_cand);
            TRIALS_HIP_TRY(hipMemcpyAsync(hT.data(), cand_T(c), (size_t)n_cand * 64, hipMemcpyDeviceToHost, c->stream));
            TRIALS_HIP_TRY(hipMemcpyAsync(hP.data(), cand_P(c), (size_t)n_cand * 64, hipMemcpyDeviceToHost, c->stream));
            TRIALS_HIP_TRY(hipMemcpyAsync(hL.data(), cand_lcp(c), (size_t)n_cand * 4, hipMemcpyDeviceToHost, c->stream));
            TRIALS_HIP_TRY(hipMemcpyAsync(hB.data(), cand_base(c), (size_t)n_cand * 4, hipMemcpyDeviceToHost, c->stream));
            TRIALS_HIP_TRY(hipStreamSynchronize(c->stream));
        }
        for (int t = 0; t < nTp; ++t) {
            stocs_trial_result& R = B->out[(size_t)(t0 + t)];
            const int b0 = c->trial_first_base[(size_t)t], b1 = c->trial_first_base[(size_t)t + 1];
            const int o0 = c->trial_cand_off[(size_t)t], o1 = c->trial_cand_off[(size_t)t + 1];
            R.n_bases = b1 - b0;
            R.n_candidates = o1 - o0;
            R.n_quads = 0;
            std::vector<long long>& q = B->quads[(size_t)(t0 + t)];
            q.resize((size_t)(b1 - b0));
            for (int b = b0; b < b1; ++b) {
                const long long nq = (size_t)b + 1 < c->quad_off.size() ? (long long)(c->quad_off[(size_t)b + 1] - c->quad_off[(size_t)b]) : 0;
                q[(size_t)(b - b0)] = nq; R.n_quads += nq;
            }
            uint32_t lo, hi;
            memcpy(&lo, &out18[(size_t)t * 18], 4); memcpy(&hi, &out18[(size_t)t * 18 + 1], 4);
            const uint64_t key = ((uint64_t)hi << 32) | lo;
            R.best_lcp = 0.0f; R.best_index = -1;
            memset(R.best_pose16, 0, sizeof(R.best_pose16));
            if (key) {
                uint32_t id = 0;
                stocs_unpack_best(key, &R.best_lcp, &id);
                R.best_index = (int32_t)id;
                memcpy(R.best_pose16, &out18[(size_t)t * 18 + 2], 64);
            }
            if (B->keep && o1 > o0) {
                B->T[(size_t)(t0 + t)].assign(hT.begin() + (size_t)o0 * 16, hT.begin() + (size_t)o1 * 16);
                B->P[(size_t)(t0 + t)].assign(hP.begin() + (size_t)o0 * 16, hP.begin() + (size_t)o1 * 16);
                B->lcp[(size_t)(t0 + t)].assign(hL.begin() + o0, hL.begin() + o1);
                B->cbase[(size_t)(t0 + t)].assign(hB.begin() + o0, hB.begin() + o1);
            }
        }
        ms_res += now_ms() - ta;
        t0 = t1;
    }
#undef TRIALS_HIP_TRY
    // the context is left as stocs_reset_trial leaves it: no bases, no candidates (the batch's results live in the batch record)
    clear_trial_batch(c);
    c->bases.clear(); c->quad_off.clear(); clear_candidates(c);
    c->best_lcp = 0; c->best_index = -1;
    if (rc) { B->nT = 0; B->out.clear(); return rc; }     // no results: stocs_trials_get_* answer STOCS_ERR_STATE ("trial out of range") for a failed batch
    {
        const double t_end = CallTiming::now_s();
        auto put = [&](const char* what, double ms) { if (TM.n < CallTiming::MAX_STEPS) { TM.label[TM.n] = what; TM.ms[TM.n] = ms; ++TM.n; } };
        put("congruent sets of all trials (stocs_find_congruent_all over the concatenated base sets, all pieces)", ms_cong);
        put("transforms of all trials (stocs_make_transforms, all pieces)", ms_xf);
        put("verification: scoring launch(es) + per-trial arg-max + read-back, all pieces", ms_ver);
        put("host: batch record set-up", ms_setup);
        put("host: base sets of the pieces assembled", ms_asm);
        put("host: per-trial results (+ candidate read-back with keep_details)", ms_res);
        put("whole call by the host clock", now_ms() - t_entry);
        put("pieces (sets of launches) the batch was cut into", (double)B->pieces);
        TM.t_last = t_end;
    }
    if (out) memcpy(out, B->out.data(), sizeof(stocs_trial_result) * (size_t)nT);
    return STOCS_OK;
}

static TrialBatch* batch_of(stocs_ctx* c, int trial) {
    if (!c || !c->trials) { set_error("no trial batch (call stocs_run_trials first)"); return NULL; }
    TrialBatch* B = (TrialBatch*)c->trials;
    if (trial < 0 || trial >= B->nT) { set_error("trial %d out of range (%d trials in the last batch)", trial, B->nT); return NULL; }
    return B;
}

int stocs_trials_get_bases(stocs_ctx* c, int trial, int32_t* base_ids4, float* inv2, int32_t* valid, int cap_attempts, int* n_attempts) {
    TrialBatch* B = batch_of(c, trial);
    if (!B) return STOCS_ERR_STATE;
    if (n_attempts) *n_attempts = B->nA;
    for (int a = 0; a < B->nA && a < cap_attempts; ++a) {
        const BaseOut& r = B->res[(size_t)trial * B->nA + a];
        if (base_ids4) for (int k = 0; k < 4; ++k) base_ids4[4 * a + k] = r.ids[k];
        if (inv2) { inv2[2 * a] = r.inv[0]; inv2[2 * a + 1] = r.inv[1]; }
        if (valid) valid[a] = r.valid;
    }
    return B->nA > cap_attempts && (base_ids4 || inv2 || valid) ? STOCS_ERR_CAPACITY : STOCS_OK;
}

int stocs_trials_get_quad_counts(stocs_ctx* c, int trial, int64_t* counts, int cap, int* n) {
    TrialBatch* B = batch_of(c, trial);
    if (!B || !n) return B ? STOCS_ERR_INVALID : STOCS_ERR_STATE;
    const std::vector<long long>& q = B->quads[(size_t)trial];
    *n = (int)q.size();
    if (counts) for (int b = 0; b < *n && b < cap; ++b) counts[b] = (int64_t)q[(size_t)b];
    return counts && *n > cap ? STOCS_ERR_CAPACITY : STOCS_OK;
}

int stocs_trials_get_candidates(stocs_ctx* c, int trial, float* T16_centred, float* pose16_camera, float* lcp, int32_t* base_index, int cap, int* n) {
    TrialBatch* B = batch_of(c, trial);
    if (!B || !n) return B ? STOCS_ERR_INVALID : STOCS_ERR_STATE;
    *n = B->out[(size_t)trial].n_candidates;
    if (!(T16_centred || pose16_camera || lcp || base_index)) return STOCS_OK;
    if (!B->keep) { set_error("stocs_trials_get_candidates: the batch was run without keep_details"); return STOCS_ERR_STATE; }
    const int m = std::min(*n, cap);
    if (m > 0) {
        if (T16_centred) memcpy(T16_centred, B->T[(size_t)trial].data(), (size_t)m * 64);
        if (pose16_camera) memcpy(pose16_camera, B->P[(size_t)trial].data(), (size_t)m * 64);
        if (lcp) memcpy(lcp, B->lcp[(size_t)trial].data(), (size_t)m * 4);
        if (base_index) memcpy(base_index, B->cbase[(size_t)trial].data(), (size_t)m * 4);
    }
    return *n > cap ? STOCS_ERR_CAPACITY : STOCS_OK;
}

}  // extern "C"

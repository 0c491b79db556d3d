// The facade serves the reference's one-call-per-attempt / per-base loops from batched GPU passes (include/stocs.hpp,
// sample_one / lookahead_slot).  This program checks that the batching is invisible: every result equals what one C-ABI
// call per reference call gives on a second, independent context -- across the block boundary of the look-ahead, with
// the sampling and the congruent-set calls interleaved, and with a base the facade did not sample in between.
//   facade_percall_check <scene.stcl> <model.stcl>        exit code 0 = all equal
#include <cstdio>
#include <cstring>
#include <iostream>

#include <stocs.hpp>

static bool read_stcl(const std::string& path, std::vector<float>& pos, std::vector<float>& nrm, std::vector<float>* prob, std::vector<int32_t>* pixel) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[8];
    int32_t n = 0, flags = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && std::string(magic, 8) == "STOCSCL1" && fread(&n, 4, 1, f) == 1 && fread(&flags, 4, 1, f) == 1 && n >= 0;
    if (ok) {
        pos.resize((size_t)n * 3); nrm.resize((size_t)n * 3);
        ok = fread(pos.data(), 4, pos.size(), f) == pos.size() && fread(nrm.data(), 4, nrm.size(), f) == nrm.size();
        if (ok && (flags & 1)) { std::vector<float> p(n); ok = fread(p.data(), 4, n, f) == (size_t)n; if (prob) *prob = p; }
        else if (prob) prob->assign(n, 1.0f);
        if (ok && (flags & 2)) { std::vector<int32_t> px((size_t)n * 2); ok = fread(px.data(), 4, px.size(), f) == px.size(); if (pixel) *pixel = px; }
    }
    fclose(f);
    return ok;
}

// one C-ABI call per reference call, on the plain context of a second estimator
struct PerCall {
    stocs_ctx* ctx;
    uint64_t seed;
    bool sample(int attempt, int32_t ids[4], float inv[2]) {
        int32_t valid = 0;
        if (stocs_sample_bases(ctx, 0, seed, attempt, 1, 0.0f, ids, inv, &valid) != STOCS_OK) { std::cerr << stocs_last_error() << std::endl; exit(3); }
        return valid != 0;
    }
    std::vector<int32_t> quads(const int32_t ids[4], const float inv[2]) {
        int64_t total = 0, n = 0;
        if (stocs_set_bases(ctx, 1, ids, inv) != STOCS_OK || stocs_find_congruent_all(ctx, &total) != STOCS_OK) { std::cerr << stocs_last_error() << std::endl; exit(3); }
        std::vector<int32_t> q((size_t)total * 4 + 4);
        if (stocs_get_quads(ctx, 0, q.data(), total, &n) != STOCS_OK) { std::cerr << stocs_last_error() << std::endl; exit(3); }
        q.resize((size_t)n * 4);
        return q;
    }
};

static bool same_quads(const std::vector<Quadrilateral>& a, const std::vector<int32_t>& b) {
    if (a.size() * 4 != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        for (int k = 0; k < 4; ++k) if (a[i][k] != b[4 * i + (size_t)k]) return false;
    return true;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::cerr << "usage: facade_percall_check <scene.stcl> <model.stcl>" << std::endl; return 2; }
    stocs::SceneCloud scene;
    stocs::ModelCloud model;
    if (!read_stcl(argv[1], scene.pos, scene.nrm, &scene.class_probability, &scene.pixel) || !read_stcl(argv[2], model.pos, model.nrm, NULL, NULL)) {
        std::cerr << "cannot read the clouds" << std::endl;
        return 2;
    }
    const uint64_t seed = 4242;
    stocs::stocs_estimator facade(model, scene, ".", 640, 480, 0.005f, 5, 5, 0.0f, 0.10f), plain(model, scene, ".", 640, 480, 0.005f, 5, 5, 0.0f, 0.10f);
    facade.set_seed(seed);
    PerCall ref{plain.context(), seed};
    int mismatches = 0, n_bases = 0, n_quads = 0;
    struct Base { std::vector<int> ids; float i1, i2; };
    std::vector<Base> bases;
    // 1. 130 attempts: one look-ahead block and a part of the next
    for (int a = 0; a < 130; ++a) {
        std::vector<int> ids(4, -1);
        float i1 = 0, i2 = 0;
        const bool ok = facade.sample_class_base(ids, i1, i2);
        int32_t rid[4]; float rinv[2];
        const bool rok = ref.sample(a, rid, rinv);
        if (ok != rok) { ++mismatches; continue; }
        if (ok) {
            for (int k = 0; k < 4; ++k) mismatches += ids[(size_t)k] != rid[k];
            mismatches += std::memcmp(&i1, &rinv[0], 4) != 0 || std::memcmp(&i2, &rinv[1], 4) != 0;
            bases.push_back(Base{ids, i1, i2});
            ++n_bases;
        }
    }
    // 2. congruent sets of the second block's bases first (they are the ones in the context), then of the first block's
    //    (no longer in the context: the per-base path), then a base nobody sampled (permuted ids), then a cached one again
    std::vector<size_t> order;
    for (size_t b = bases.size(); b-- > 0;) order.push_back(b);
    for (size_t t = 0; t < order.size() && t < 40; ++t) {
        Base& B = bases[order[t]];
        std::vector<Quadrilateral> q;
        facade.find_congruent_sets_on_model(B.ids, B.i1, B.i2, &q);
        const int32_t rid[4] = {B.ids[0], B.ids[1], B.ids[2], B.ids[3]};
        const float rinv[2] = {B.i1, B.i2};
        const std::vector<int32_t> rq = ref.quads(rid, rinv);
        mismatches += !same_quads(q, rq);
        n_quads += (int)q.size();
        if (t == 5 || t == 17) {   // a base the facade has not seen, in the middle of the sequence
            std::vector<int> odd = {B.ids[1], B.ids[0], B.ids[3], B.ids[2]};
            std::vector<Quadrilateral> q2;
            facade.find_congruent_sets_on_model(odd, B.i2, B.i1, &q2);
            const int32_t oid[4] = {odd[0], odd[1], odd[2], odd[3]};
            const float oinv[2] = {B.i2, B.i1};
            mismatches += !same_quads(q2, ref.quads(oid, oinv));
        }
    }
    // 3. sampling and searching interleaved from a fresh seed
    facade.set_seed(seed + 1);
    ref.seed = seed + 1;
    for (int a = 0; a < 12; ++a) {
        std::vector<int> ids(4, -1);
        float i1 = 0, i2 = 0;
        const bool ok = facade.sample_class_base(ids, i1, i2);
        int32_t rid[4]; float rinv[2];
        const bool rok = ref.sample(a, rid, rinv);
        mismatches += ok != rok;
        if (!ok || !rok) continue;
        for (int k = 0; k < 4; ++k) mismatches += ids[(size_t)k] != rid[k];
        std::vector<Quadrilateral> q;
        facade.find_congruent_sets_on_model(ids, i1, i2, &q);
        mismatches += !same_quads(q, ref.quads(rid, rinv));
        n_quads += (int)q.size();
    }
    std::cout << "facade_percall_check: bases " << n_bases << ", congruent sets compared " << n_quads << ", mismatches " << mismatches << std::endl;
    return mismatches == 0 && n_bases > 20 ? 0 : 1;
}

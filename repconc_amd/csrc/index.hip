// Stateful PQ index behind the C ABI (SURVEY.md §8b proposal: rc_index_create / set_centroids / add_codes / search).
//
// Replaces what the reference keeps inside Faiss objects: `initialize_index` + `add_docs`
// (models/repconc/evaluate_repconc.py:78-98: IndexPQ(D, M, 8, IP), centroids <- C.ravel(), codes appended raw) and
// `index.search` (:182, models/jpq/finetune_jpq.py:176).  The library owns the device memory: codes [ntotal, M] uint8
// in one growing allocation (capacity doubles, so appending the corpus chunk by chunk is amortised O(N) — the reference's
// add_docs round-trips the whole code vector through numpy on every call), the [M,256,dsub] centroid table (rewritten in
// place by rc_index_set_centroids: the JPQ per-step `synchronize_model_index`, finetune_jpq.py:209-214, is a 786 KB
// copy), and the search workspace.  rc_index_search is the host loop around rc_adc_search: it reads the status word
// and retries with another selection slack exactly like repconc_amd.ops.adc_search.  Python uses torch-owned tensors
// (repconc_amd/index.py) over the stateless entry points; this file is for hosts without torch.
#include "rc_common.h"
#include "adc_common.h"

struct rc_index_s {
    rc_handle_t h;
    int D, M, K;
    float* C;            // [M, K, D/M]
    uint8_t* codes;      // [cap, M]
    uint8_t* image;      // [cap, M] permuted copy for the conflict-free ADC screen (NULL when this M uses none)
    int64_t n, cap;
    void* ws;
    size_t ws_bytes;
    int* status;         // device word
    int* qstatus;        // per-query status words of the last search, grown on demand (no allocation on the search path)
    int qstatus_cap;
    bool have_centroids;
};

#define RC_IDX_HIP(idx, call)                                                  \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) { (idx)->h->last_hip_error = (int)e_; return RC_EHIP; } \
    } while (0)

extern "C" int rc_index_create(rc_handle_t h, int D, int M, int K, rc_index_t* out) {
    rc_device_guard device_guard_(h);
    if (!h || !out || D <= 0 || M <= 0) return RC_EINVAL;
    if (K != RC_K || D % M != 0) return RC_ESHAPE;
    rc_index_s* idx = new (std::nothrow) rc_index_s();
    if (!idx) return RC_EINVAL;
    idx->h = h; idx->D = D; idx->M = M; idx->K = K;
    idx->C = nullptr; idx->codes = nullptr; idx->image = nullptr; idx->n = 0; idx->cap = 0; idx->ws = nullptr; idx->ws_bytes = 0;
    idx->status = nullptr; idx->have_centroids = false;
    hipError_t e = hipMalloc((void**)&idx->C, (size_t)M * K * (D / M) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&idx->status, 256);
    if (e != hipSuccess) {
        h->last_hip_error = (int)e;
        if (idx->C) (void)hipFree(idx->C);
        delete idx;
        return RC_EHIP;
    }
    *out = idx;
    return RC_OK;
}

extern "C" int rc_index_destroy(rc_index_t idx) {
    rc_device_guard device_guard_(idx ? idx->h : nullptr);
    if (!idx) return RC_EINVAL;
    if (idx->C) (void)hipFree(idx->C);
    if (idx->codes) (void)hipFree(idx->codes);
    if (idx->image) (void)hipFree(idx->image);
    if (idx->ws) (void)hipFree(idx->ws);
    if (idx->status) (void)hipFree(idx->status);
    if (idx->qstatus) (void)hipFree(idx->qstatus);
    delete idx;
    return RC_OK;
}

extern "C" int64_t rc_index_ntotal(rc_index_t idx) { return idx ? idx->n : -1; }
extern "C" const uint8_t* rc_index_codes(rc_index_t idx) { return idx ? idx->codes : nullptr; }
extern "C" const float* rc_index_centroids(rc_index_t idx) { return (idx && idx->have_centroids) ? idx->C : nullptr; }

extern "C" int rc_index_set_centroids(rc_index_t idx, const float* C, rc_stream_t stream) {
    rc_device_guard device_guard_(idx ? idx->h : nullptr);
    if (!idx || !C) return RC_EINVAL;
    RC_IDX_HIP(idx, hipMemcpyAsync(idx->C, C, (size_t)idx->M * idx->K * (idx->D / idx->M) * sizeof(float),
                                   hipMemcpyDeviceToDevice, (hipStream_t)stream));
    idx->have_centroids = true;
    return RC_OK;
}

extern "C" int rc_index_reserve(rc_index_t idx, int64_t rows, rc_stream_t stream) {
    rc_device_guard device_guard_(idx ? idx->h : nullptr);
    if (!idx || rows < 0) return RC_EINVAL;
    if (rows <= idx->cap) return RC_OK;
    if (rows > 0xFFFFFFFFll) return RC_ESHAPE;
    uint8_t* fresh = nullptr;
    uint8_t* fresh_img = nullptr;
    const bool with_image = rc_adc_scan_image_bytes(1, idx->M) > 0;
    RC_IDX_HIP(idx, hipMalloc((void**)&fresh, (size_t)rows * idx->M));
    if (with_image) {
        hipError_t e = hipMalloc((void**)&fresh_img, rc_adc_scan_image_bytes(rows, idx->M));
        if (e != hipSuccess) { (void)hipFree(fresh); idx->h->last_hip_error = (int)e; return RC_EHIP; }
    }
    hipStream_t s = (hipStream_t)stream;
    if (idx->n > 0) {
        hipError_t e = hipMemcpyAsync(fresh, idx->codes, (size_t)idx->n * idx->M, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess && with_image)
            e = hipMemcpyAsync(fresh_img, idx->image, rc_adc_scan_image_bytes(idx->n, idx->M), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);            // the old blocks are freed right below
        if (e != hipSuccess) {
            (void)hipFree(fresh);
            if (fresh_img) (void)hipFree(fresh_img);
            idx->h->last_hip_error = (int)e;
            return RC_EHIP;
        }
    }
    if (idx->codes) (void)hipFree(idx->codes);
    if (idx->image) (void)hipFree(idx->image);
    idx->codes = fresh;
    idx->image = fresh_img;
    idx->cap = rows;
    return RC_OK;
}

extern "C" int rc_index_add_codes(rc_index_t idx, const uint8_t* codes, int64_t n, rc_stream_t stream) {
    rc_device_guard device_guard_(idx ? idx->h : nullptr);
    if (!idx || n < 0 || (n > 0 && !codes)) return RC_EINVAL;
    if (n == 0) return RC_OK;
    if (idx->n + n > idx->cap) {
        int64_t want = idx->cap * 2;
        if (want < idx->n + n) want = idx->n + n;
        if (want > 0xFFFFFFFFll) want = 0xFFFFFFFFll;
        if (want < idx->n + n) return RC_ESHAPE;
        const int rc = rc_index_reserve(idx, want, stream);
        if (rc != RC_OK) return rc;
    }
    RC_IDX_HIP(idx, hipMemcpyAsync(idx->codes + (size_t)idx->n * idx->M, codes, (size_t)n * idx->M,
                                   hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (idx->image) {                                    // keep the permuted image in step, row for row
        const int rc = rc_adc_scan_image(idx->h, idx->codes, idx->n, n, idx->M, idx->image, stream);
        if (rc != RC_OK) return rc;
    }
    idx->n += n;
    return RC_OK;
}

extern "C" int rc_index_reset(rc_index_t idx) {
    rc_device_guard device_guard_(idx ? idx->h : nullptr);
    if (!idx) return RC_EINVAL;
    idx->n = 0;
    return RC_OK;
}

// Synchronous (reads the status word between attempts).  scores [nq,k], ids [nq,k] on the device.
extern "C" int rc_index_search(rc_index_t idx, const float* q, int nq, int k, float* scores, int64_t* ids,
                               rc_stream_t stream) {
    rc_device_guard device_guard_(idx ? idx->h : nullptr);
    if (!idx || nq < 0 || k <= 0 || (nq > 0 && (!q || !scores || !ids))) return RC_EINVAL;
    if (!idx->have_centroids) return RC_EINVAL;
    if (nq == 0) return RC_OK;
    hipStream_t s = (hipStream_t)stream;
    if (idx->n == 0) {                                   // Faiss semantics: -inf scores, -1 labels
        const size_t cnt = (size_t)nq * k;
        float* hs = (float*)malloc(cnt * sizeof(float));
        int64_t* hi = (int64_t*)malloc(cnt * sizeof(int64_t));
        if (!hs || !hi) { free(hs); free(hi); return RC_EINVAL; }
        for (size_t i = 0; i < cnt; ++i) { hs[i] = -INFINITY; hi[i] = -1; }
        hipError_t e = hipMemcpyAsync(scores, hs, cnt * sizeof(float), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(ids, hi, cnt * sizeof(int64_t), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        free(hs); free(hi);
        if (e != hipSuccess) { idx->h->last_hip_error = (int)e; return RC_EHIP; }
        return RC_OK;
    }
    if (!adc_search_supported(idx->M)) {
        // a width without a screening kernel (Faiss's IndexPQ takes any M: evaluate_repconc.py:81): the exact scan answers
        const size_t need_x = rc_adc_search_exact_ws_bytes(idx->n, idx->M, idx->K, nq, k);
        if (need_x == 0) return RC_ESHAPE;
        if (need_x > idx->ws_bytes) {
            if (idx->ws) (void)hipFree(idx->ws);
            idx->ws = nullptr; idx->ws_bytes = 0;
            RC_IDX_HIP(idx, hipMalloc(&idx->ws, need_x));
            idx->ws_bytes = need_x;
        }
        const int rcx = rc_adc_search_exact(idx->h, idx->codes, idx->n, idx->M, idx->K, idx->C, idx->D, q, nq, k, 0, scores, ids,
                                            idx->ws, idx->ws_bytes, stream);
        if (rcx != RC_OK) return rcx;
        RC_IDX_HIP(idx, hipStreamSynchronize(s));
        return RC_OK;
    }
    const size_t need = idx->image ? rc_adc_search_img_ws_bytes(idx->n, idx->M, idx->K, nq, k)
                                   : rc_adc_search_ws_bytes(idx->n, idx->M, idx->K, nq, k);
    if (need == 0) return RC_ESHAPE;
    if (need > idx->ws_bytes) {
        if (idx->ws) (void)hipFree(idx->ws);
        idx->ws = nullptr; idx->ws_bytes = 0;
        RC_IDX_HIP(idx, hipMalloc(&idx->ws, need));
        idx->ws_bytes = need;
    }
    // Like Faiss's IndexPQ.search (evaluate_repconc.py:180-185) this returns for any index content: the sampled-threshold
    // search is tried twice on the whole batch; queries whose status bits are still set then go through the exact path
    // (rc_adc_search_exact: no sample, no threshold), alone — the other queries' results stand.
    // kept by the index, grown on demand: hipMalloc / hipFree per search cost an allocator round trip and a device-wide
    // synchronisation on the small-batch path (JPQ steps, validation)
    if (nq > idx->qstatus_cap) {
        if (idx->qstatus) (void)hipFree(idx->qstatus);
        idx->qstatus = nullptr; idx->qstatus_cap = 0;
        const int cap = nq < 1024 ? 1024 : nq;
        RC_IDX_HIP(idx, hipMalloc((void**)&idx->qstatus, (size_t)cap * sizeof(int)));
        idx->qstatus_cap = cap;
    }
    int* qstatus = idx->qstatus;
    struct guard { void* p[4]; ~guard() { for (void* q_ : p) if (q_) (void)hipFree(q_); } } g = {{nullptr, nullptr, nullptr, nullptr}};
    double slack = 6.0;
    int st = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        RC_IDX_HIP(idx, hipMemsetAsync(idx->status, 0, sizeof(int), s));
        RC_IDX_HIP(idx, hipMemsetAsync(qstatus, 0, (size_t)nq * sizeof(int), s));
        const int rc = rc_adc_search_q(idx->h, idx->codes, idx->image, idx->n, idx->M, idx->K, idx->C, idx->D, q, nq, k, 0,
                                       slack, scores, ids, idx->status, qstatus, idx->ws, idx->ws_bytes, stream);
        if (rc != RC_OK) return rc;
        RC_IDX_HIP(idx, hipMemcpyAsync(&st, idx->status, sizeof(int), hipMemcpyDeviceToHost, s));
        RC_IDX_HIP(idx, hipStreamSynchronize(s));
        if (st == 0) return RC_OK;
        slack = (st & 1) ? slack * 3.0 + 2.0 : (slack / 3.0);     // too few candidates -> widen; overflow -> tighten
    }
    int* hq = (int*)malloc((size_t)nq * sizeof(int));
    if (!hq) return RC_EINVAL;
    hipError_t e = hipMemcpy(hq, qstatus, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost);
    int nbad = 0;
    for (int i = 0; e == hipSuccess && i < nq; ++i)
        if (hq[i]) hq[nbad++] = i;                               // compacted in place: indices of the failing queries
    if (e != hipSuccess) { free(hq); idx->h->last_hip_error = (int)e; return RC_EHIP; }
    float* bq = nullptr; float* bs = nullptr; int64_t* bi = nullptr;
    const size_t need_x = rc_adc_search_exact_ws_bytes(idx->n, idx->M, idx->K, nbad, k);
    if (e == hipSuccess) e = hipMalloc((void**)&bq, (size_t)nbad * idx->D * sizeof(float));
    g.p[1] = bq;
    if (e == hipSuccess) e = hipMalloc((void**)&bs, (size_t)nbad * k * sizeof(float));
    g.p[2] = bs;
    if (e == hipSuccess) e = hipMalloc((void**)&bi, (size_t)nbad * k * sizeof(int64_t));
    g.p[3] = bi;
    if (e == hipSuccess && need_x > idx->ws_bytes) {
        (void)hipFree(idx->ws);
        idx->ws = nullptr; idx->ws_bytes = 0;
        e = hipMalloc(&idx->ws, need_x);
        if (e == hipSuccess) idx->ws_bytes = need_x;
    }
    for (int j = 0; e == hipSuccess && j < nbad; ++j)
        e = hipMemcpyAsync(bq + (size_t)j * idx->D, q + (size_t)hq[j] * idx->D, (size_t)idx->D * sizeof(float), hipMemcpyDeviceToDevice, s);
    int rc = RC_OK;
    if (e == hipSuccess)
        rc = rc_adc_search_exact(idx->h, idx->codes, idx->n, idx->M, idx->K, idx->C, idx->D, bq, nbad, k, 0, bs, bi, idx->ws,
                                 idx->ws_bytes, stream);
    for (int j = 0; e == hipSuccess && rc == RC_OK && j < nbad; ++j) {
        e = hipMemcpyAsync(scores + (size_t)hq[j] * k, bs + (size_t)j * k, (size_t)k * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess)
            e = hipMemcpyAsync(ids + (size_t)hq[j] * k, bi + (size_t)j * k, (size_t)k * sizeof(int64_t), hipMemcpyDeviceToDevice, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    free(hq);
    if (e != hipSuccess) { idx->h->last_hip_error = (int)e; return RC_EHIP; }
    return rc;
}

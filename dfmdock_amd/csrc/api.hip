// api.hip - C ABI of the engine (include/dfmdock_amd.h): handles, weight packing, the per-evaluation
// kernel schedule and the Euler-Maruyama loop.  Everything runs on one HIP stream per complex handle;
// the host only enqueues (no sync inside the 40-step loop).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "dfm_device.h"
#include "dfm_internal.h"

using namespace dfm;

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
extern "C" const char *dfm_last_error(void) { return g_err.c_str(); }

static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            return fail(_e == hipErrorOutOfMemory ? DFM_E_OOM : DFM_E_HIP,                         \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                       \
        }                                                                                         \
    } while (0)

// Every handle belongs to one device (dfm_model: the device current at creation; dfm_complex: its model's).  Entry points
// run under a DeviceScope: switch to the handle's device, restore the caller's on the way out.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};
#define DEVICE_SCOPE(dev)                                                                          \
    DeviceScope _ds(dev);                                                                          \
    if (_ds.err != hipSuccess) return fail(DFM_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(_ds.err))

// ------------------------------------------------------------------------------------------------
// Device blocks released by one handle and wanted by the next: a set driver creates and destroys a complex (about forty buffers, some of them
// gigabytes) every few hundred milliseconds, and hipMalloc / hipFree of that size cost milliseconds each (hipFree also drains the device).
// Released blocks are kept per device, up to a quarter of its memory, and handed out again to requests of at most twice-smaller size;
// DFM_ALLOC_CACHE=0 turns the cache off.
struct BlockCache {
    std::mutex m;
    std::multimap<size_t, void *> free_blocks[MAX_DEVICES];
    size_t bytes[MAX_DEVICES] = {};
    size_t cap[MAX_DEVICES] = {};
    static bool enabled()
    {
        static const bool on = [] { const char *e = getenv("DFM_ALLOC_CACHE"); return !(e && atoi(e) == 0); }();
        return on;
    }
    static double fraction()      // share of a device's memory the cache may park (DFM_ALLOC_CACHE_FRAC, default 0.25)
    {
        static const double f = [] {
            const char *e = getenv("DFM_ALLOC_CACHE_FRAC");
            const double v = e ? atof(e) : 0.25;
            return v < 0.0 ? 0.0 : (v > 0.9 ? 0.9 : v);
        }();
        return f;
    }
    bool give(int dev, void *p, size_t size)
    {
        std::lock_guard<std::mutex> g(m);
        if (!cap[dev]) {
            size_t fr = 0, tot = 0;
            DeviceScope ds(dev);      // the memory of the block's OWN device, whatever the calling thread's current device is
            if (ds.err != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) tot = 0;
            cap[dev] = (size_t)((double)tot * fraction()) + 1;
        }
        if (bytes[dev] + size > cap[dev]) return false;
        free_blocks[dev].emplace(size, p);
        bytes[dev] += size;
        return true;
    }
    // hand every parked block of `dev` (all devices: dev < 0) back to the driver; returns the bytes freed
    size_t trim(int dev)
    {
        std::vector<std::pair<int, void *>> drop;
        size_t freed = 0;
        {
            std::lock_guard<std::mutex> g(m);
            for (int d = 0; d < MAX_DEVICES; ++d) {
                if (dev >= 0 && d != dev) continue;
                for (auto &kv : free_blocks[d]) drop.emplace_back(d, kv.second);
                freed += bytes[d];
                free_blocks[d].clear();
                bytes[d] = 0;
            }
        }
        for (auto &dp : drop) { DeviceScope ds(dp.first); (void)hipFree(dp.second); }
        return freed;
    }
};
static BlockCache &g_block_cache = *new BlockCache;      // never destroyed: a handle may outlive static destruction at process exit

// diagnostic: DFM_ALLOC_GUARD=<KiB> puts that many KiB of 0xA5 before and after every block (cache off) and checks them at release:
// a kernel writing outside its buffers is reported on stderr with the block's size and the first damaged offset
static size_t guard_bytes()
{
    static const size_t g = [] { const char *e = getenv("DFM_ALLOC_GUARD"); return e ? (size_t)atoi(e) * 1024 : (size_t)0; }();
    return g;
}
struct DevPool {
    struct Block { void *p; size_t size; int dev; };
    std::vector<Block> ptrs;
    hipStream_t owner = nullptr;      // bind(): the one stream that ever touches this pool's blocks
    bool bound = false;
    void bind(hipStream_t s) { owner = s; bound = true; }
    ~DevPool() { release(); }
    // `drained`: the caller has synchronised the ONE stream that ever touched these blocks (a complex handle's own stream), so
    // nothing in flight reads them and the device-wide wait - which would also wait for every OTHER handle's queued work, e.g. a
    // whole dfm_sample call of the next complex of a set run - is not needed.
    void release(bool drained = false)
    {
        if (ptrs.empty()) return;
        if (const size_t G = guard_bytes()) {
            (void)hipDeviceSynchronize();
            std::vector<unsigned char> h(G);
            for (const Block &b : ptrs) {
                unsigned char *base = reinterpret_cast<unsigned char *>(b.p) - G;
                for (int side = 0; side < 2; ++side) {
                    (void)hipMemcpy(h.data(), side ? base + G + b.size : base, G, hipMemcpyDeviceToHost);
                    for (size_t k = 0; k < G; ++k)
                        if (h[k] != 0xA5) {
                            fprintf(stderr, "DFM_ALLOC_GUARD: block of %zu bytes: %s guard damaged at offset %zu (byte 0x%02x)\n", b.size,
                                    side ? "TAIL" : "HEAD", k, h[k]);
                            break;
                        }
                }
                (void)hipFree(base);
            }
            ptrs.clear();
            return;
        }
        if (BlockCache::enabled()) {
            // what hipFree would have done: nothing in flight reads these blocks when the next owner gets them (a bound pool waits
            // for its own stream only)
            if (!drained) { if (bound) (void)hipStreamSynchronize(owner); else (void)hipDeviceSynchronize(); }
            for (const Block &b : ptrs)
                if (b.dev < 0 || b.dev >= MAX_DEVICES || !g_block_cache.give(b.dev, b.p, b.size)) (void)hipFree(b.p);
        } else {
            for (const Block &b : ptrs) (void)hipFree(b.p);
        }
        ptrs.clear();
    }
    // hand back the blocks allocated after `mark` (= ptrs.size() before a group of allocations that failed half way); the caller has
    // synchronised the owning stream
    void release_tail(size_t mark)
    {
        if (mark >= ptrs.size()) return;
        std::vector<Block> tail(ptrs.begin() + mark, ptrs.end()), head(ptrs.begin(), ptrs.begin() + mark);
        ptrs.swap(tail);
        release(true);
        ptrs.swap(head);
    }
    template <typename T> hipError_t alloc(T **out, size_t n)
    {
        size_t bytes = (n ? n : 1) * sizeof(T);
        int dev = -1;
        (void)hipGetDevice(&dev);
        void *p = nullptr;
        if (const size_t G = guard_bytes()) {
            unsigned char *base = nullptr;
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&base), bytes + 2 * G);
            if (e != hipSuccess) return e;
            (void)hipMemset(base, 0xA5, G); (void)hipMemset(base + G + bytes, 0xA5, G);
            (void)hipDeviceSynchronize();
            ptrs.push_back({base + G, bytes, dev});
            *out = reinterpret_cast<T *>(base + G);
            return hipSuccess;
        }
        if (BlockCache::enabled() && dev >= 0 && dev < MAX_DEVICES) {
            bytes = (bytes + 65535) & ~(size_t)65535;      // 64 KiB granules: neighbouring sizes share blocks
            // the block's true size travels with it: look it up by taking from the cache under the lock
            {
                std::lock_guard<std::mutex> g(g_block_cache.m);
                auto &fb = g_block_cache.free_blocks[dev];
                auto it = fb.lower_bound(bytes);
                if (it != fb.end() && it->first <= 2 * bytes + (1u << 20)) {
                    p = it->second;
                    bytes = it->first;
                    g_block_cache.bytes[dev] -= it->first;
                    fb.erase(it);
                }
            }
        }
        if (!p) {
            hipError_t e = hipMalloc(&p, bytes);
            if (e != hipSuccess && BlockCache::enabled() && dev >= 0 && dev < MAX_DEVICES) {      // out of memory with blocks parked in the cache: drop them and retry
                std::vector<void *> drop;
                {
                    std::lock_guard<std::mutex> g(g_block_cache.m);
                    for (auto &kv : g_block_cache.free_blocks[dev]) drop.push_back(kv.second);
                    g_block_cache.free_blocks[dev].clear();
                    g_block_cache.bytes[dev] = 0;
                }
                for (void *q : drop) (void)hipFree(q);
                (void)hipGetLastError();
                e = hipMalloc(&p, bytes);
            }
            if (e != hipSuccess) return e;
        }
        ptrs.push_back({p, bytes, dev});
        *out = reinterpret_cast<T *>(p);
        // diagnostic: DFM_ALLOC_POISON=<byte> fills every block handed out (fresh or from the cache) with that byte - 255 = NaN
        // patterns in fp32 / fp16 - so that a kernel reading memory nobody wrote shows up as a changed or non-finite result
        static const int poison = [] { const char *e = getenv("DFM_ALLOC_POISON"); return e ? atoi(e) & 255 : -1; }();
        if (poison >= 0) {
            hipError_t e = bound ? hipMemsetAsync(p, poison, bytes, owner) : hipMemset(p, poison, bytes);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    template <typename T> hipError_t upload(T **out, const T *host, size_t n)
    {
        hipError_t e = alloc(out, n);
        if (e != hipSuccess) return e;
        return hipMemcpy(*out, host, n * sizeof(T), hipMemcpyHostToDevice);
    }
    // the same on a handle's own (non-blocking) stream: the caller synchronises it before `host` may change
    template <typename T> hipError_t upload_async(T **out, const T *host, size_t n, hipStream_t s)
    {
        hipError_t e = alloc(out, n);
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(*out, host, n * sizeof(T), hipMemcpyHostToDevice, s);
    }
};

struct dfm_model {
    dfm_hparams hp;
    int device = 0;
    DevPool pool;
    float *single_embed = nullptr;   // [256][lm]
    LayerDev layers[8];
    HeadsDev heads;
    float *en0_w = nullptr;          // [256][512]
    PairHeadDev pair[3];             // family 1: 0 to_force, 1 to_energy, 2 to_confidence
    PairHeadDev dist;                // family 1: to_dist (fp32 only; w3 = [256][64], transposed)
    float *ir0_w = nullptr, *ir0_b = nullptr, *ir2_w = nullptr, *ir2_b = nullptr, *ir4_w = nullptr, *ir4_b = nullptr;   // to_ires
    float tab_max[8][2] = {};        // per layer: largest |entry| of the two merged lookup tables as stored (log2e-scaled; before the fp16 clamp)
};

struct Workspace {
    int Bcap = 0;
    DevPool pool;
    float4 *pos = nullptr, *ca4 = nullptr, *cb4 = nullptr;      // centred backbone N / CA / virtual CB of every trajectory, [B][N] each
    int32_t *edges = nullptr; uint32_t *codes = nullptr; float *radial = nullptr;
    float *h = nullptr, *h2 = nullptr, *A = nullptr, *Bm = nullptr, *agg = nullptr, *u = nullptr;
    uint16_t *Bmb = nullptr, *mbuf = nullptr;
    float *gn_shift = nullptr, *gn_den = nullptr, *gn_part = nullptr;
    float *fvec = nullptr, *en_part = nullptr; int32_t *clash_part = nullptr;
    float *fpart = nullptr, *cpart = nullptr, *conf = nullptr;    // family 1: force partials per receptor tile, confidence
    float *pair_s = nullptr;                                      // family 1, 16-bit engines: s(r, l) of one pair head, [B][L][32 ceil(R / 32)]
    float *scores = nullptr, *lig_cur = nullptr, *tr_update = nullptr, *rot_update = nullptr, *t_dev = nullptr;
    float *hid_base = nullptr;       // [max(Bcap, 1024)][2][128]: time-dependent half of the score-scale MLPs (k_time_embed), per trajectory
                                     // (dfm_score) or per step of the time grid (dfm_sample)
    float *ir1 = nullptr, *ir2 = nullptr, *ir3 = nullptr;      // to_ires scratch, allocated on the first DFM_F_IRES call
    // layer 0 behind the message table (allocated on first use): per edge the source of its gated message, the row list of the
    // edges the edge model still evaluates, their messages, the list's length and the running total for the profile
    uint32_t *task_ctr = nullptr;      // per-workgroup task + exit counters of the 16-bit message kernel (dynamic tasks, kernels_edge.hip)
    uint32_t *l0_src = nullptr; uint4 *l0_rows = nullptr; uint16_t *l0_x = nullptr; uint32_t *l0_counter = nullptr;
    float *l0_x32 = nullptr;      // fp32 engine: the row list's messages (1 KiB per row)
    unsigned long long *l0_miss_total = nullptr;
    // replayed step graph: {evaluations started, seed lo, seed hi, -} and the per-step scalars of the call's time grid
    uint32_t *step_ctl = nullptr; StepParams *step_params = nullptr;
    // t_dev / hid_base / step_params live in their own pool, sized for max(Bcap, the longest time grid asked for so far)
    DevPool tpool;
    size_t Tcap = 0;
};

struct dfm_complex {
    dfm_model *m = nullptr;
    int device = 0;
    int homomer = 0;                 // value of the 67th position channel (positional_embed_dim 67)
    int R = 0, L = 0, N = 0, K = 0, knn = 0, nsamp = 0;
    DevPool pool;
    float *rec_pos = nullptr, *lig0 = nullptr;
    float *h0 = nullptr, *A0 = nullptr, *Bm0 = nullptr;
    float *A0s = nullptr; uint16_t *Bmb0 = nullptr;   // SILU_S * A0 (fp32), SILU_S * Bm0 (fp16): layer-0 operands of the 16-bit engines
    uint16_t *A0h = nullptr;                          // SILU_S * A0 as fp16 (bf16-operand message kernel)
    Workspace ws;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;      // profiling events (pairs)
    std::vector<char> ev_lig;        // per pair: the launch covered the ligand nodes only
    size_t ev_used = 0;
    hipEvent_t ev_total[2] = {nullptr, nullptr};
    dfm_profile prof = {};
    unsigned long long *stamp_dev = nullptr;   // [8 waves][4 phases] (diagnostic builds)
    uint32_t fwd_counter = 0;
    // layer-0 message table of the 16-bit engine (kernels_edge.hip: k_l0_gather): gated messages of every intra-chain ordered pair
    // [R*R + L*L][256] fp16 and the feature code each entry was built with; rebuilt after set_pose / set_homomer
    DevPool l0_pool, l0_pool32;
    uint16_t *l0_table = nullptr; uint2 *l0_code0 = nullptr;
    bool l0_valid = false;
    float *l0_table32 = nullptr; uint2 *l0_code0_32 = nullptr; bool l0_valid32 = false;      // the fp32 engine's table (1 KiB per pair)
    std::vector<hipEvent_t> ev_l0;   // profiling events of the table path (triples: before rows | between | after gather)
    size_t ev_l0_used = 0;
    // One step of dfm_sample (score evaluation + heads + Euler-Maruyama update [+ clash force]) captured as a hipGraph and replayed
    // num_steps times: every per-step / per-call scalar is read from device memory (ws.step_ctl, ws.step_params), so the nodes are
    // the same in every step AND every call - the executable graph is kept until the buffers it points into move
    hipGraphExec_t step_exec = nullptr;
    uint64_t step_key = 0;       // (B, engine flags) the graph was captured for
    uint64_t buf_gen = 0;        // bumped whenever a buffer a captured launch points into is re-allocated (workspace, message table)
    uint64_t step_gen = ~0ull;   // buf_gen at capture
};

// ------------------------------------------------------------------------------------------------
extern "C" void dfm_default_hparams(dfm_hparams *hp)
{
    hp->lm_embed_dim = 1301; hp->positional_embed_dim = 66; hp->spatial_embed_dim = 100;
    hp->node_dim = 256; hp->edge_dim = 128; hp->inner_dim = 128; hp->depth = 6; hp->knn = 20; hp->n_sample = 40;
    hp->cut_off = 20.0f; hp->mask_dist = 22.0f;
    hp->r3_min_sigma = 0.1; hp->r3_max_sigma = 30.0; hp->so3_min_sigma = 0.1; hp->so3_max_sigma = 1.5;
    hp->family = 0; hp->agg_mean = 1;
}

struct BlobMap {
    const float *single_embed, *spatial_embed, *positional_embed;
    struct Lw {
        const float *e1_w, *e1_b, *e2_w, *e2_b, *n1_w, *n1_b, *gn_w, *gn_b, *gn_ms, *n2_w, *n2_b, *c1_w, *c1_b, *c2_w,
            *att_w, *att_b;
    } layer[8];
    const float *en0_w, *en_ln_w, *en_ln_b, *en3_w;
    struct Ph { const float *w0, *ln_w, *ln_b, *w3; } pair[3], dist;   // family 1: to_force, to_energy, to_confidence; to_dist (w3 is [64][256])
    const float *ir0_w, *ir0_b, *ir2_w, *ir2_b, *ir4_w, *ir4_b;   // to_ires
    const float *t_W, *t_lin, *trs0, *trs_ln_w, *trs_ln_b, *trs4, *rots0, *rots_ln_w, *rots_ln_b, *rots4;
    int64_t total;
};

static void map_blob(const dfm_hparams *hp, const float *blob, BlobMap *w)
{   // state_dict order of Score_Net (score_net_mlsb.py:249-341, egnn.py:37-93) or, family 1, of EGNN_Net
    // (egnn_net.py:296-385); see dfmdock_amd/weights.py
    const int64_t Hh = hp->node_dim, He = hp->edge_dim, Hi = hp->inner_dim;
    const float *p = blob;
    auto take = [&](const float *&dst, int64_t n) { dst = p; p += n; };
    take(w->single_embed, Hh * hp->lm_embed_dim);
    take(w->spatial_embed, He * hp->spatial_embed_dim);
    take(w->positional_embed, He * hp->positional_embed_dim);
    for (int l = 0; l < hp->depth; ++l) {
        auto &Lw = w->layer[l];
        take(Lw.e1_w, Hh * (2 * Hh + 1 + He)); take(Lw.e1_b, Hh);
        take(Lw.e2_w, Hh * Hh); take(Lw.e2_b, Hh);
        take(Lw.n1_w, Hh * 2 * Hh); take(Lw.n1_b, Hh);
        take(Lw.gn_w, Hh); take(Lw.gn_b, Hh); take(Lw.gn_ms, Hh);
        take(Lw.n2_w, Hh * Hh); take(Lw.n2_b, Hh);
        if (l == hp->depth - 1 && hp->family == 0) { take(Lw.c1_w, Hh * Hh); take(Lw.c1_b, Hh); take(Lw.c2_w, Hh); }
        else Lw.c1_w = Lw.c1_b = Lw.c2_w = nullptr;
        take(Lw.att_w, Hh); take(Lw.att_b, 1);
    }
    if (hp->family == 1) {   // to_energy, to_force, to_dist, to_confidence on cat[h_r, h_l, D]
        auto head = [&](BlobMap::Ph &h) { take(h.w0, Hh * (2 * Hh + 1)); take(h.ln_w, Hh); take(h.ln_b, Hh); take(h.w3, Hh); };
        head(w->pair[1]); head(w->pair[0]);
        take(w->dist.w0, Hh * (2 * Hh + 1)); take(w->dist.ln_w, Hh); take(w->dist.ln_b, Hh); take(w->dist.w3, 64 * Hh);
        head(w->pair[2]);
        w->en0_w = w->en_ln_w = w->en_ln_b = w->en3_w = nullptr;
    } else {
        take(w->en0_w, Hh * 2 * Hh); take(w->en_ln_w, Hh); take(w->en_ln_b, Hh); take(w->en3_w, Hh);
        for (auto &h : w->pair) h.w0 = h.ln_w = h.ln_b = h.w3 = nullptr;
        w->dist.w0 = w->dist.ln_w = w->dist.ln_b = w->dist.w3 = nullptr;
    }
    take(w->ir0_w, 2 * Hh * Hh); take(w->ir0_b, 2 * Hh); take(w->ir2_w, 4 * Hh * Hh); take(w->ir2_b, 2 * Hh);   // to_ires.{0,2}
    take(w->ir4_w, 2 * Hh); take(w->ir4_b, 1);                                                                  // to_ires.4
    take(w->t_W, Hi / 2); take(w->t_lin, Hi * Hi);
    take(w->trs0, Hi * (Hi + 1)); take(w->trs_ln_w, Hi); take(w->trs_ln_b, Hi); take(w->trs4, Hi);
    take(w->rots0, Hi * (Hi + 1)); take(w->rots_ln_w, Hi); take(w->rots_ln_b, Hi); take(w->rots4, Hi);
    w->total = (int64_t)(p - blob);
}

extern "C" int64_t dfm_param_count(const dfm_hparams *hp)
{
    if (!hp || hp->depth < 1 || hp->depth > 8 || hp->family < 0 || hp->family > 1) return -1;
    BlobMap w;
    map_blob(hp, nullptr, &w);
    return w.total;
}

extern "C" int dfm_device_count(int *count)
{
    if (!count) return fail(DFM_E_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(DFM_E_NODEVICE, hipGetErrorString(e)); }
    *count = n;
    return DFM_OK;
}

extern "C" int dfm_set_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(DFM_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(DFM_E_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DFM_E_NODEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    return DFM_OK;
}

static int degree_of(const dfm_hparams &hp, int N, int *knn, int *ns)
{   // score_net_mlsb.py:89-94
    int k = hp.knn, s = hp.n_sample;
    if (N < k) { k = N; s = 0; }
    if (N < k + s) s = N - k;
    *knn = k; *ns = s;
    return k + s;
}

// ------------------------------------------------------------------------------------------------
extern "C" int dfm_diffusion_coef(const dfm_hparams *hp, int which, double t, double *g_out, double *sigma_out)
{
    if (!hp) return fail(DFM_E_INVALID, "hp is NULL");
    double sigma, g;
    if (which == 0) {          // r3_diffuser.py:20-24
        sigma = hp->r3_min_sigma * std::pow(hp->r3_max_sigma / hp->r3_min_sigma, t);
        g = sigma * std::sqrt(2.0 * (std::log(hp->r3_max_sigma) - std::log(hp->r3_min_sigma)));
    } else if (which == 1) {   // so3_diffuser.py:210-227
        if (t < 0.0 || t > 1.0 || t != t) return fail(DFM_E_INVALID, "Invalid t (so3 sigma needs 0 <= t <= 1)");
        sigma = std::log(t * std::exp(hp->so3_max_sigma) + (1.0 - t) * std::exp(hp->so3_min_sigma));
        g = std::sqrt(2.0 * (std::exp(hp->so3_max_sigma) - std::exp(hp->so3_min_sigma)) * sigma / std::exp(sigma));
    } else {
        return fail(DFM_E_INVALID, "which must be 0 (R^3) or 1 (SO(3))");
    }
    if (g_out) *g_out = g;
    if (sigma_out) *sigma_out = sigma;
    return DFM_OK;
}

// ------------------------------------------------------------------------------------------------
static std::vector<uint16_t> pack_frags(const float *W /*[256 out][256 in]*/, bool f16 = false)
{   // 16-bit B-operand fragments of v_mfma_f32_32x32x16_bf16: [kk][nt][lane][e] = W[nt*32 + lane%32][chan(kk, lane/32, e)]
    std::vector<uint16_t> f((size_t)16 * 8 * 64 * 8);
    for (int kk = 0; kk < 16; ++kk)
        for (int nt = 0; nt < 8; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int n = nt * 32 + (lane & 31), k = frag_channel(kk, lane >> 5, e);
                    f[(((size_t)kk * 8 + nt) * 64 + lane) * 8 + e] = f16 ? f2h(W[(size_t)n * H + k]) : f2bf(W[(size_t)n * H + k]);
                }
    return f;
}
// split-bf16 operands of k_gemm_split, tiled in K-stage order: [K/32][4 k-groups][Nout][8] (W is [Nout][K] row-major)
static void split_bf16(const float *W, int Nout, int K, std::vector<uint16_t> &hi, std::vector<uint16_t> &lo)
{
    hi.resize((size_t)Nout * K); lo.resize((size_t)Nout * K);
    for (int o = 0; o < Nout; ++o)
        for (int k = 0; k < K; ++k) {
            const float w = W[(size_t)o * K + k];
            const size_t d = (((size_t)(k / 32) * 4 + (k % 32) / 8) * Nout + o) * 8 + (k % 8);
            hi[d] = f2bf(w);
            lo[d] = f2bf(w - bf2f(hi[d]));
        }
}
// SILU_S * bias as packed (hi | lo << 16) 16-bit pairs: the B operand of the bias k-step of the MFMA edge kernels
static std::vector<uint32_t> pack_bias(const float *bias, bool f16)
{
    std::vector<uint32_t> v(2 * H, 0u);      // [8 n-tiles][64 lanes]: lanes 0..31 = columns, lanes 32..63 = 0 (k >= 8)
    for (int c = 0; c < H; ++c) {
        const float x = SILU_S * bias[c];
        uint16_t hi, lo;
        if (f16) { hi = f2h(x); lo = f2h(x - h2f(hi)); }
        else { hi = f2bf(x); lo = f2bf(x - bf2f(hi)); }
        v[(c / 32) * 64 + (c % 32)] = (uint32_t)hi | ((uint32_t)lo << 16);
    }
    return v;
}
static std::vector<float> transpose256(const float *W)
{
    std::vector<float> t((size_t)H * H);
    for (int o = 0; o < H; ++o)
        for (int k = 0; k < H; ++k) t[(size_t)k * H + o] = W[(size_t)o * H + k];
    return t;
}

extern "C" dfm_model *dfm_model_create(const float *blob, size_t n_floats, const dfm_hparams *hp)
{
    if (!blob || !hp) { fail(DFM_E_INVALID, "blob or hp is NULL"); return nullptr; }
    if (hp->node_dim != H || hp->edge_dim != HE || hp->inner_dim != HI || hp->spatial_embed_dim != 100 ||
        (hp->positional_embed_dim != 66 && hp->positional_embed_dim != 67) || hp->depth < 1 || hp->depth > 8 || hp->knn < 1 || hp->n_sample < 0 ||
        hp->knn + hp->n_sample > 60 || hp->lm_embed_dim < 1 || hp->family < 0 || hp->family > 1) {
        fail(DFM_E_INVALID, "unsupported hyper-parameters (kernels are built for node 256 / edge 128 / inner 128, degree <= 60)");
        return nullptr;
    }
    BlobMap w;
    map_blob(hp, blob, &w);
    if ((int64_t)n_floats != w.total) {
        fail(DFM_E_INVALID, "blob has " + std::to_string(n_floats) + " floats, expected " + std::to_string(w.total));
        return nullptr;
    }
    dfm_model *m = new dfm_model();
    m->hp = *hp;
    if (hipGetDevice(&m->device) != hipSuccess) { fail(DFM_E_NODEVICE, "no current HIP device"); delete m; return nullptr; }
    const int Pd = hp->positional_embed_dim;
    DevPool &P = m->pool;
    bool ok = true;
    auto up = [&](float **dst, const float *src, size_t n) { ok = ok && P.upload(dst, src, n) == hipSuccess; };
    auto up16 = [&](uint16_t **dst, const std::vector<uint16_t> &v) { ok = ok && P.upload(dst, v.data(), v.size()) == hipSuccess; };
    auto up32 = [&](uint32_t **dst, const std::vector<uint32_t> &v) { ok = ok && P.upload(dst, v.data(), v.size()) == hipSuccess; };
    const int Kin1 = 2 * H + 1 + HE;
    up(&m->single_embed, w.single_embed, (size_t)H * hp->lm_embed_dim);
    for (int l = 0; l < hp->depth && ok; ++l) {
        const auto &Lw = w.layer[l];
        LayerDev &D = m->layers[l];
        std::memset(&D, 0, sizeof(D));
        std::vector<float> Wab((size_t)2 * H * H), bias_ab(2 * H, 0.f), w_r(H);
        for (int c = 0; c < H; ++c) {
            std::memcpy(&Wab[(size_t)c * H], Lw.e1_w + (size_t)c * Kin1, H * sizeof(float));
            std::memcpy(&Wab[(size_t)(H + c) * H], Lw.e1_w + (size_t)c * Kin1 + H, H * sizeof(float));
            bias_ab[c] = Lw.e1_b[c];
            w_r[c] = Lw.e1_w[(size_t)c * Kin1 + 2 * H];
        }
        // T[idx][c] = sum_k We[c][k] * SP[k][idx]  (one_hot @ spatial/positional_embed^T then We: a row gather)
        std::vector<double> Td((size_t)NTAB * H);
        std::vector<float> T((size_t)NTAB * H);
        for (int idx = 0; idx < NTAB; ++idx)
            for (int c = 0; c < H; ++c) {
                double s = 0;
                for (int k = 0; k < HE; ++k) {
                    const float sp = idx < 100 ? w.spatial_embed[(size_t)k * 100 + idx]
                                               : w.positional_embed[(size_t)k * Pd + (idx - 100)];
                    s += (double)Lw.e1_w[(size_t)c * Kin1 + 2 * H + 1 + k] * sp;
                }
                Td[(size_t)idx * H + c] = s;
                T[(size_t)idx * H + c] = (float)s;
            }
        // merged fp16 tables of the 16-bit MFMA kernel, pre-multiplied by SILU_S like every other SiLU input there
        const double S = (double)SILU_S;
        std::vector<uint16_t> T2b((size_t)NTAB2 * H);
        double tmax[2] = {0, 0};      // range telemetry of dfm_complex_selfcheck
#if DFM_TAB_MERGE
        for (int om = 0; om < 24; ++om)
            for (int th = 0; th < 24; ++th)
                for (int ph = 0; ph < 12; ++ph) {
                    uint16_t *row = &T2b[(size_t)((om * 24 + th) * 12 + ph) * H];
                    const double *a = &Td[(size_t)(40 + om) * H], *b = &Td[(size_t)(64 + th) * H], *c3 = &Td[(size_t)(88 + ph) * H];
                    for (int c = 0; c < H; ++c) { const double v = S * (a[c] + b[c] + c3[c]); tmax[0] = std::fmax(tmax[0], std::fabs(v)); row[c] = f2h((float)v); }
                }
        for (int rp = 0; rp < 66; ++rp)
            for (int d = 0; d < 40; ++d) {
                uint16_t *row = &T2b[(size_t)(6912 + rp * 40 + d) * H];
                const double *a = &Td[(size_t)(100 + rp) * H], *b = &Td[(size_t)d * H];
                for (int c = 0; c < H; ++c) { const double v = S * (a[c] + b[c]); tmax[1] = std::fmax(tmax[1], std::fabs(v)); row[c] = f2h((float)v); }
            }
        m->tab_max[l][0] = (float)tmax[0]; m->tab_max[l][1] = (float)tmax[1];
#else
        for (int c = 0; c < H; ++c) {
            for (int om = 0; om < 24; ++om)
                for (int th = 0; th < 24; ++th)
                    T2b[(size_t)(om * 24 + th) * H + c] = f2h((float)(S * (Td[(size_t)(40 + om) * H + c] + Td[(size_t)(64 + th) * H + c])));
            for (int ph = 0; ph < 12; ++ph)
                for (int d = 0; d < 40; ++d)
                    T2b[(size_t)(576 + ph * 40 + d) * H + c] = f2h((float)(S * (Td[(size_t)(88 + ph) * H + c] + Td[(size_t)d * H + c])));
            for (int rp = 0; rp < 66; ++rp) T2b[(size_t)(1056 + rp) * H + c] = f2h((float)(S * Td[(size_t)(100 + rp) * H + c]));
        }
#endif
        {
            std::vector<float> wrs(H), babs(2 * H), wc2s(H);
            for (int c = 0; c < H; ++c) wrs[c] = SILU_S * w_r[c];
            for (int c = 0; c < 2 * H; ++c) babs[c] = SILU_S * bias_ab[c];
            up(&D.w_r_s, wrs.data(), H); up(&D.bias_ab_s, babs.data(), 2 * H);
            up32(&D.b2p, pack_bias(Lw.e2_b, false)); up32(&D.b2p16, pack_bias(Lw.e2_b, true));
            if (Lw.c1_w) {
                for (int c = 0; c < H; ++c) wc2s[c] = Lw.c2_w[c] / SILU_S;
                up(&D.wc2_s, wc2s.data(), H);
                up32(&D.bc1p, pack_bias(Lw.c1_b, false)); up32(&D.bc1p16, pack_bias(Lw.c1_b, true));
            }
        }
        if (Pd == 67) {   // "sym" channel: one_hot-free constant column of the position matrix -> We . P[:, 66] on every edge
            std::vector<float> bh(bias_ab), bhs(2 * H);
            for (int c = 0; c < H; ++c) {
                double s2 = 0;
                for (int k = 0; k < HE; ++k) s2 += (double)Lw.e1_w[(size_t)c * Kin1 + 2 * H + 1 + k] * w.positional_embed[(size_t)k * Pd + 66];
                bh[c] = (float)((double)bias_ab[c] + s2);
            }
            for (int c = 0; c < 2 * H; ++c) bhs[c] = SILU_S * bh[c];
            up(&D.bias_ab_h, bh.data(), 2 * H); up(&D.bias_ab_h_s, bhs.data(), 2 * H);
        }
        up(&D.Wab, Wab.data(), Wab.size()); up(&D.bias_ab, bias_ab.data(), bias_ab.size());
        up(&D.w_r, w_r.data(), w_r.size()); up(&D.T, T.data(), T.size()); up16(&D.T2b, T2b);
        const std::vector<float> W2t = transpose256(Lw.e2_w);
        up(&D.W2t, W2t.data(), W2t.size()); up16(&D.W2f, pack_frags(Lw.e2_w)); up16(&D.W2f16, pack_frags(Lw.e2_w, true));
        up(&D.b2, Lw.e2_b, H); up(&D.att_w, Lw.att_w, H); D.att_b = Lw.att_b[0];
        up(&D.W3, Lw.n1_w, (size_t)H * 2 * H); up(&D.b3, Lw.n1_b, H);
        up(&D.gn_w, Lw.gn_w, H); up(&D.gn_b, Lw.gn_b, H); up(&D.gn_ms, Lw.gn_ms, H);
        up(&D.W4, Lw.n2_w, (size_t)H * H); up(&D.b4, Lw.n2_b, H);
        {
            std::vector<uint16_t> hi, lo;
            std::vector<float> Wabs(Wab.size());      // [Wa|Wb] of the 16-bit engine: scaled by SILU_S (A, Bm feed a SiLU)
            for (size_t q = 0; q < Wab.size(); ++q) Wabs[q] = SILU_S * Wab[q];
            split_bf16(Wabs.data(), 2 * H, H, hi, lo); up16(&D.Wab_hi, hi); up16(&D.Wab_lo, lo);
            split_bf16(Lw.n1_w, H, 2 * H, hi, lo); up16(&D.W3_hi, hi); up16(&D.W3_lo, lo);
            split_bf16(Lw.n2_w, H, H, hi, lo); up16(&D.W4_hi, hi); up16(&D.W4_lo, lo);
        }
        if (Lw.c1_w) {
            const std::vector<float> Wc1t = transpose256(Lw.c1_w);
            up(&D.Wc1t, Wc1t.data(), Wc1t.size()); up16(&D.Wc1f, pack_frags(Lw.c1_w)); up16(&D.Wc1f16, pack_frags(Lw.c1_w, true));
            up(&D.bc1, Lw.c1_b, H); up(&D.wc2, Lw.c2_w, H);
        }
    }
    HeadsDev &Hd = m->heads;
    std::memset(&Hd, 0, sizeof(Hd));
    std::memset(m->pair, 0, sizeof(m->pair));
    if (hp->family == 1) {
        const int Kp = 2 * H + 1;
        for (int q = 0; q < 3 && ok; ++q) {
            const auto &src = w.pair[q];
            std::vector<float> wab((size_t)2 * H * H), wd(H);
            for (int c = 0; c < H; ++c) {      // rows 0..255: receptor half W[:, :256]; rows 256..511: ligand half W[:, 256:512]
                std::memcpy(&wab[(size_t)c * H], src.w0 + (size_t)c * Kp, H * sizeof(float));
                std::memcpy(&wab[(size_t)(H + c) * H], src.w0 + (size_t)c * Kp + H, H * sizeof(float));
                wd[c] = src.w0[(size_t)c * Kp + 2 * H];
            }
            PairHeadDev &D = m->pair[q];
            std::vector<uint16_t> hi, lo;
            up(&D.wab, wab.data(), wab.size());
            split_bf16(wab.data(), 2 * H, H, hi, lo); up16(&D.wab_hi, hi); up16(&D.wab_lo, lo);
            up(&D.w_d, wd.data(), H); up(&D.ln_w, src.ln_w, H); up(&D.ln_b, src.ln_b, H); up(&D.w3, src.w3, H);
        }
        if (ok) {      // to_dist: fp32 projection only (evaluated on request, DFM_F_DIST), Linear(256 -> 64) stored transposed
            const auto &src = w.dist;
            std::vector<float> wab((size_t)2 * H * H), wd(H), w3t((size_t)H * 64);
            for (int c = 0; c < H; ++c) {
                std::memcpy(&wab[(size_t)c * H], src.w0 + (size_t)c * Kp, H * sizeof(float));
                std::memcpy(&wab[(size_t)(H + c) * H], src.w0 + (size_t)c * Kp + H, H * sizeof(float));
                wd[c] = src.w0[(size_t)c * Kp + 2 * H];
                for (int o = 0; o < 64; ++o) w3t[(size_t)c * 64 + o] = src.w3[(size_t)o * H + c];
            }
            PairHeadDev &D = m->dist;
            std::memset(&D, 0, sizeof(D));
            up(&D.wab, wab.data(), wab.size()); up(&D.w_d, wd.data(), H); up(&D.ln_w, src.ln_w, H); up(&D.ln_b, src.ln_b, H);
            up(&D.w3, w3t.data(), w3t.size());
        }
    } else {
        up(&m->en0_w, w.en0_w, (size_t)H * 2 * H);
        Hd.en_wa = m->en0_w; Hd.en_wb = m->en0_w ? m->en0_w + H : nullptr;
        up(&Hd.en_ln_w, w.en_ln_w, H); up(&Hd.en_ln_b, w.en_ln_b, H); up(&Hd.en_w3, w.en3_w, H);
    }
    up(&m->ir0_w, w.ir0_w, (size_t)2 * H * H); up(&m->ir0_b, w.ir0_b, 2 * H); up(&m->ir2_w, w.ir2_w, (size_t)4 * H * H);
    up(&m->ir2_b, w.ir2_b, 2 * H); up(&m->ir4_w, w.ir4_w, 2 * H); up(&m->ir4_b, w.ir4_b, 1);
    up(&Hd.t_W, w.t_W, HI / 2); up(&Hd.t_lin, w.t_lin, (size_t)HI * HI);
    up(&Hd.trs0, w.trs0, (size_t)HI * (HI + 1)); up(&Hd.trs_ln_w, w.trs_ln_w, HI); up(&Hd.trs_ln_b, w.trs_ln_b, HI);
    up(&Hd.trs4, w.trs4, HI);
    up(&Hd.rots0, w.rots0, (size_t)HI * (HI + 1)); up(&Hd.rots_ln_w, w.rots_ln_w, HI); up(&Hd.rots_ln_b, w.rots_ln_b, HI);
    up(&Hd.rots4, w.rots4, HI);
    if (!ok) {
        fail(DFM_E_HIP, std::string("weight upload failed: ") + hipGetErrorString(hipGetLastError()));
        delete m;
        return nullptr;
    }
    return m;
}

extern "C" void dfm_model_destroy(dfm_model *m)
{
    if (!m) return;
    DeviceScope ds(m->device);
    delete m;
}

// ------------------------------------------------------------------------------------------------
// layer-0 operands of the 16-bit MFMA edge kernel: SILU_S * A0 (fp32 and fp16) and SILU_S * Bm0 (fp16)
__global__ void k_scale_ab(const float *__restrict__ A0, const float *__restrict__ Bm0, float *__restrict__ A0s,
                           uint16_t *__restrict__ A0h, uint16_t *__restrict__ Bmb0, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { A0s[i] = SILU_S * A0[i]; A0h[i] = f2h(SILU_S * A0[i]); Bmb0[i] = f2h(SILU_S * Bm0[i]); }
}

// layer 0's [Wa|Wb] projection of the node embedding: pose independent, once per complex (again when the homomer flag changes)
static hipError_t project_layer0(dfm_complex *cx)
{
    const dfm_model *m = cx->m;
    const LayerDev &L0 = m->layers[0];
    const int N = cx->N;
    GemmArgs g;
    std::memset(&g, 0, sizeof(g));
    g.A0 = cx->h0; g.lda = H; g.K = H; g.W = L0.Wab; g.ldw = H; g.bias = cx->homomer ? L0.bias_ab_h : L0.bias_ab; g.M = N;
    g.Nout = 2 * H; g.epi = 2; g.C = cx->A0; g.ldc = H; g.C2 = cx->Bm0;
    hipError_t e = launch_gemm_f32(g, cx->stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_scale_ab, dim3((N * H + 255) / 256), dim3(256), token_lds(), cx->stream, cx->A0, cx->Bm0, cx->A0s, cx->A0h, cx->Bmb0, N * H);
    return hipGetLastError();
}

extern "C" dfm_complex *dfm_complex_create(dfm_model *m, const float *rec_x, const float *lig_x, const float *rec_pos,
                                           const float *lig_pos, int R, int L)
{
    if (!m || !rec_x || !lig_x || !rec_pos || !lig_pos) { fail(DFM_E_INVALID, "NULL argument"); return nullptr; }
    if (R < 1 || L < 1 || R + L > MAX_NODES) { fail(DFM_E_INVALID, "need 1 <= R, L and R + L <= 4096"); return nullptr; }
    DeviceScope ds(m->device);
    if (ds.err != hipSuccess) { fail(DFM_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(ds.err)); return nullptr; }
    dfm_complex *cx = new dfm_complex();
    cx->m = m; cx->device = m->device; cx->R = R; cx->L = L; cx->N = R + L;
    cx->K = degree_of(m->hp, cx->N, &cx->knn, &cx->nsamp);
    const int N = cx->N, lm = m->hp.lm_embed_dim;
    DevPool &P = cx->pool;
    // A handle's stream is NON-BLOCKING and everything the handle does - uploads included - is ordered on it: nothing here goes
    // through the legacy default stream, whose implicit synchronisation would make the creation of the next complex of a set run
    // wait for the whole dfm_sample call of the current one (driver.run_set overlaps the two from separate host threads).
    bool ok = hipStreamCreateWithFlags(&cx->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreate(&cx->ev_total[0]) == hipSuccess && hipEventCreate(&cx->ev_total[1]) == hipSuccess;
    float *x = nullptr;
    cx->pool.bind(cx->stream); cx->ws.pool.bind(cx->stream); cx->l0_pool.bind(cx->stream); cx->l0_pool32.bind(cx->stream);
    DevPool tmp;      // staging of the raw node features: a cached block (hipMalloc / hipFree would drain the device)
    tmp.bind(cx->stream);
    ok = ok && tmp.alloc(&x, (size_t)N * lm) == hipSuccess;
    ok = ok && hipMemcpyAsync(x, rec_x, (size_t)R * lm * sizeof(float), hipMemcpyHostToDevice, cx->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(x + (size_t)R * lm, lig_x, (size_t)L * lm * sizeof(float), hipMemcpyHostToDevice, cx->stream) == hipSuccess;
    ok = ok && P.upload_async(&cx->rec_pos, rec_pos, (size_t)R * 9, cx->stream) == hipSuccess;
    ok = ok && P.upload_async(&cx->lig0, lig_pos, (size_t)L * 9, cx->stream) == hipSuccess;
    ok = ok && P.alloc(&cx->h0, (size_t)N * H) == hipSuccess && P.alloc(&cx->A0, (size_t)N * H) == hipSuccess;
    ok = ok && P.alloc(&cx->Bm0, (size_t)N * H) == hipSuccess && P.alloc(&cx->Bmb0, (size_t)N * H) == hipSuccess;
    ok = ok && P.alloc(&cx->A0s, (size_t)N * H) == hipSuccess;
    ok = ok && P.alloc(&cx->A0h, (size_t)N * H) == hipSuccess;
#if defined(DFM_EDGE_STAMP) || defined(DFM_EDGE_TRACE)
    ok = ok && P.alloc(&cx->stamp_dev, 48 + 8 * 130) == hipSuccess && hipMemsetAsync(cx->stamp_dev, 0, (48 + 8 * 130) * 8, cx->stream) == hipSuccess;
#else      // product: four words for the message kernel's clock stamps (profiled calls only, dfm_profile::edge_shader_cycles)
    ok = ok && P.alloc(&cx->stamp_dev, 8) == hipSuccess && hipMemsetAsync(cx->stamp_dev, 0, 8 * 8, cx->stream) == hipSuccess;
#endif
    if (ok) {
        // node = single_embed(cat[rec_x, lig_x]) (score_net_mlsb.py:365-366): pose independent, once per complex
        GemmArgs g;
        std::memset(&g, 0, sizeof(g));
        g.A0 = x; g.lda = lm; g.K = lm; g.W = m->single_embed; g.ldw = lm; g.M = N; g.Nout = H; g.C = cx->h0; g.ldc = H;
        ok = launch_gemm_f32(g, cx->stream) == hipSuccess;
        ok = ok && project_layer0(cx) == hipSuccess;
    }
    // (also on failure: the uploads above may still be reading the caller's buffers)
    const bool drained = cx->stream && hipStreamSynchronize(cx->stream) == hipSuccess;
    ok = ok && drained;
    tmp.release(drained);
    if (!ok) {
        fail(DFM_E_HIP, std::string("complex creation failed: ") + hipGetErrorString(hipGetLastError()));
        dfm_complex_destroy(cx);
        return nullptr;
    }
    return cx;
}

extern "C" void dfm_complex_destroy(dfm_complex *cx)
{
    if (!cx) return;
    DeviceScope ds(cx->device);
    bool drained = false;
    if (cx->stream) drained = hipStreamSynchronize(cx->stream) == hipSuccess;
    // every block of the handle was only ever touched by its own stream, which is drained now: no device-wide wait
    cx->ws.pool.release(drained); cx->ws.tpool.release(drained); cx->l0_pool.release(drained); cx->l0_pool32.release(drained); cx->pool.release(drained);
    if (cx->stream) (void)hipStreamDestroy(cx->stream);
    if (cx->step_exec) (void)hipGraphExecDestroy(cx->step_exec);
    for (hipEvent_t e : cx->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : cx->ev_l0) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) if (cx->ev_total[i]) (void)hipEventDestroy(cx->ev_total[i]);
    delete cx;
}

extern "C" int dfm_complex_set_pose(dfm_complex *cx, const float *rec_pos, const float *lig_pos)
{
    if (!cx) return fail(DFM_E_INVALID, "NULL argument");
    DEVICE_SCOPE(cx->device);
    // the stream may still read the old poses (an earlier call never returns before its work is done, but stay safe)
    HIPCHK(hipStreamSynchronize(cx->stream));
    if (rec_pos) HIPCHK(hipMemcpyAsync(cx->rec_pos, rec_pos, (size_t)cx->R * 9 * sizeof(float), hipMemcpyHostToDevice, cx->stream));
    if (lig_pos) HIPCHK(hipMemcpyAsync(cx->lig0, lig_pos, (size_t)cx->L * 9 * sizeof(float), hipMemcpyHostToDevice, cx->stream));
    HIPCHK(hipStreamSynchronize(cx->stream));
    cx->l0_valid = false; cx->l0_valid32 = false;      // the intra-chain geometry may have changed: the layer-0 message tables are rebuilt on next use
    return DFM_OK;
}

extern "C" int dfm_complex_set_homomer(dfm_complex *cx, int flag)
{
    if (!cx) return fail(DFM_E_INVALID, "NULL argument");
    flag = flag ? 1 : 0;
    if (flag && cx->m->hp.positional_embed_dim != 67)
        return fail(DFM_E_INVALID, "the model has no sym channel (positional_embed_dim is 66)");
    if (flag == cx->homomer) return DFM_OK;
    DEVICE_SCOPE(cx->device);
    // the flag is committed only once the layer-0 projections that depend on it are rebuilt: a failed call leaves the old state
    // usable and a retry with the same flag does the work again instead of returning DFM_OK on stale operands
    const int old = cx->homomer;
    cx->homomer = flag;
    cx->l0_valid = false; cx->l0_valid32 = false;      // A0 carries the flag's bias: the layer-0 message tables are rebuilt on next use
    hipError_t e = project_layer0(cx);
    if (e == hipSuccess) e = hipStreamSynchronize(cx->stream);
    if (e != hipSuccess) {
        cx->homomer = old;
        (void)project_layer0(cx);                      // best effort: put the old projection back
        (void)hipStreamSynchronize(cx->stream);
        return fail(DFM_E_HIP, hipGetErrorString(e));
    }
    return DFM_OK;
}

extern "C" int dfm_complex_degree(const dfm_complex *cx) { return cx ? cx->K : -1; }

extern "C" long long dfm_trim_cache(int device)
{
    if (device >= MAX_DEVICES) return 0;      // no such device: nothing parked there (a negative index means every device)
    return (long long)g_block_cache.trim(device < 0 ? -1 : device);
}

// ------------------------------------------------------------------------------------------------
// Layer 0 behind the message table: budgets.  The table costs 516 B per intra-chain ordered pair (8 M pairs = 4.1 GB: 2000 + 2000
// residues); beyond it layer 0 is evaluated directly (DFM_L0_TABLE=0 in the environment: always).  Eligibility is a property of the
// COMPLEX alone (r04 also switched to the direct evaluation above 32 M edges per batched evaluation, so a trajectory's bits
// changed with the batch size at B = 890 for 300+300): the per-batch buffers cost 532 B per edge of a batched evaluation (19 MB per
// trajectory at 300+300, next to 14 MB of workspace) for every batch the 2^31-byte guard of ensure_workspace admits; a batch whose
// buffers do not fit fails with DFM_E_OOM - split it, or pass DFM_F_NO_L0_TABLE - instead of silently changing the arithmetic.
constexpr long long L0_MAX_PAIRS = 8ll << 20;
static bool l0_eligible(const dfm_complex *cx, int /*B*/)
{
    static const bool env_off = [] { const char *e = getenv("DFM_L0_TABLE"); return e && atoi(e) == 0; }();
    if (env_off || cx->m->hp.depth < 2) return false;      // (a depth-1 model's first layer is its last: it stores messages for the coordinate MLP)
    const long long pairs = (long long)cx->R * cx->R + (long long)cx->L * cx->L;
    return pairs <= L0_MAX_PAIRS;
}

constexpr int MIN_TIME_GRID = 4096;      // smallest time grid the workspace is sized for (dfm_sample grows it for longer schedules)
// t_dev [n], hid_base [n][2][128], step_params [n]: one entry per trajectory (dfm_score) or per step of the time grid (dfm_sample).
// The reference's sampler takes any num_steps (inference_base.py:403-404: np.linspace(1, eps, num_steps)), so does this one.
static int ensure_time_grid(dfm_complex *cx, size_t n)
{
    Workspace &W = cx->ws;
    if (n < (size_t)MIN_TIME_GRID) n = MIN_TIME_GRID;
    if (n <= W.Tcap) return DFM_OK;
    HIPCHK(hipStreamSynchronize(cx->stream));
    W.tpool.release(true);
    W.tpool.bind(cx->stream);
    W.Tcap = 0;
    cx->buf_gen++;
    HIPCHK(W.tpool.alloc(&W.t_dev, n)); HIPCHK(W.tpool.alloc(&W.hid_base, n * 2 * HI)); HIPCHK(W.tpool.alloc(&W.step_params, n));
    W.Tcap = n;
    return DFM_OK;
}
static int ensure_workspace(dfm_complex *cx, int B, bool bf16, bool l0 = false)
{
    Workspace &W = cx->ws;
    // the 16-bit message kernel addresses edges / codes / radial through buffer descriptors rooted at the whole [B][N][K] arrays with
    // 32-bit byte offsets: past 2 GiB the loads would silently return 0
    if ((unsigned long long)B * (unsigned long long)cx->N * (unsigned long long)cx->K * 4ull >= (1ull << 31))
        return fail(DFM_E_INVALID, "B * N * K * 4 must stay below 2^31: split the batch");
    const bool wants_mbuf = bf16 && cx->m->hp.family == 0;    // gated messages for the coordinate MLP (family 0 only)
    const bool need_mbuf = wants_mbuf && !W.mbuf;
    const bool need_l0 = l0 && (!W.l0_src || (bf16 ? !W.l0_x : !W.l0_x32));
    if (B <= W.Bcap && !need_mbuf && !need_l0) return DFM_OK;
    if (B > W.Bcap) {
        HIPCHK(hipStreamSynchronize(cx->stream));
        W.pool.release(true); W.tpool.release(true);
        W = Workspace();
        W.pool.bind(cx->stream);
        cx->buf_gen++;
        const size_t N = cx->N, L = cx->L, R = cx->R, K = cx->K, b = B;
        HIPCHK(W.pool.alloc(&W.pos, b * N)); HIPCHK(W.pool.alloc(&W.ca4, b * N)); HIPCHK(W.pool.alloc(&W.cb4, b * N));
        HIPCHK(W.pool.alloc(&W.edges, b * N * K)); HIPCHK(W.pool.alloc(&W.codes, b * N * K));
        HIPCHK(W.pool.alloc(&W.radial, b * N * K));
        HIPCHK(W.pool.alloc(&W.h, b * N * H)); HIPCHK(W.pool.alloc(&W.h2, b * N * H));
        HIPCHK(W.pool.alloc(&W.A, b * N * H)); HIPCHK(W.pool.alloc(&W.Bm, b * N * H));
        HIPCHK(W.pool.alloc(&W.Bmb, b * N * H)); HIPCHK(W.pool.alloc(&W.agg, b * N * H));
        HIPCHK(W.pool.alloc(&W.u, b * N * H));
        HIPCHK(W.pool.alloc(&W.gn_shift, b * H)); HIPCHK(W.pool.alloc(&W.gn_den, b * H));
        HIPCHK(W.pool.alloc(&W.gn_part, b * ((N + 31) / 32) * H * 2));      // (mean, M2) per 32-row half of a GEMM tile, trajectory and channel
        const size_t RT = (R + 63) / 64, NP = R > 4 * RT ? R : 4 * RT;     // partial-sum slots per trajectory
        HIPCHK(W.pool.alloc(&W.fvec, b * L * 3)); HIPCHK(W.pool.alloc(&W.en_part, b * NP * 2));
        HIPCHK(W.pool.alloc(&W.clash_part, b * NP)); HIPCHK(W.pool.alloc(&W.scores, b * 8));
        if (cx->m->hp.family == 1) {
            HIPCHK(W.pool.alloc(&W.fpart, b * RT * L * 3)); HIPCHK(W.pool.alloc(&W.cpart, b * RT * 4 * 2));
            HIPCHK(W.pool.alloc(&W.conf, b));
            HIPCHK(W.pool.alloc(&W.pair_s, b * L * (((R + 31) / 32) * 32)));
        }
        HIPCHK(W.pool.alloc(&W.lig_cur, b * L * 9)); HIPCHK(W.pool.alloc(&W.tr_update, b * 3));
        HIPCHK(W.pool.alloc(&W.rot_update, b * 3));
        HIPCHK(W.pool.alloc(&W.step_ctl, 4));
        HIPCHK(W.pool.alloc(&W.task_ctr, 2 * TASK_CTR_WGS));
        HIPCHK(hipMemsetAsync(W.task_ctr, 0, 2 * TASK_CTR_WGS * sizeof(uint32_t), cx->stream));
        { const int rc = ensure_time_grid(cx, b); if (rc) return rc; }
        W.Bcap = B;
    }
    if (wants_mbuf && !W.mbuf) { HIPCHK(W.pool.alloc(&W.mbuf, (size_t)W.Bcap * cx->L * KPAD * H)); cx->buf_gen++; }
    if (l0) {
        // the layer-0 buffers of this batch capacity: all of them or none.  A batch whose row buffers do not fit does not
        // leave a half-allocated workspace behind, and the error says how to run it anyway.
        const size_t cap = (size_t)W.Bcap * cx->N * cx->K;      // every edge of a batched evaluation may miss the table
        const size_t mark = W.pool.ptrs.size();
        const bool had_src = W.l0_src != nullptr, had_x = W.l0_x != nullptr, had_x32 = W.l0_x32 != nullptr;
        hipError_t e = hipSuccess;
        if (!had_src) {
            cx->buf_gen++;
            e = W.pool.alloc(&W.l0_src, cap);
            if (e == hipSuccess) e = W.pool.alloc(&W.l0_rows, cap);
            if (e == hipSuccess) e = W.pool.alloc(&W.l0_counter, 1);
            if (e == hipSuccess) e = W.pool.alloc(&W.l0_miss_total, 1);
            if (e == hipSuccess) e = hipMemsetAsync(W.l0_counter, 0, sizeof(uint32_t), cx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(W.l0_miss_total, 0, sizeof(unsigned long long), cx->stream);
        }
        if (e == hipSuccess && bf16 && !had_x) { cx->buf_gen++; e = W.pool.alloc(&W.l0_x, (cap + 32) * H); }      // the last tile of the row list stores all of its 32 rows
        if (e == hipSuccess && !bf16 && !had_x32) { cx->buf_gen++; e = W.pool.alloc(&W.l0_x32, cap * H); }         // fp32 engine: only the list's own rows are stored
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(cx->stream);
            W.pool.release_tail(mark);
            if (!had_src) { W.l0_src = nullptr; W.l0_rows = nullptr; W.l0_counter = nullptr; W.l0_miss_total = nullptr; }
            if (!had_x) W.l0_x = nullptr;
            if (!had_x32) W.l0_x32 = nullptr;
            char msg[256];
            snprintf(msg, sizeof(msg), "layer-0 message table: the row buffers of a batch of %d (%zu B per edge, %.1f GB) do not fit (%s): split the "
                     "batch or evaluate layer 0 directly (DFM_F_NO_L0_TABLE / leave DFM_F_L0_TABLE out)", W.Bcap, bf16 ? (size_t)532 : (size_t)1044,
                     (double)cap * (bf16 ? 532.0 : 1044.0) / 1e9, hipGetErrorString(e));
            return fail(e == hipErrorOutOfMemory ? DFM_E_OOM : DFM_E_HIP, msg);
        }
    }
    return DFM_OK;
}

// operands of layer 0's message kernel in the shipped 16-bit plan: the complex's own, pose-independent A0 / Bm0 (no batch stride)
static EdgeArgs layer0_edge_args(const dfm_complex *cx)
{
    EdgeArgs e;
    std::memset(&e, 0, sizeof(e));
    e.A = cx->A0s; e.Bm = cx->Bm0; e.Bmb = cx->Bmb0; e.Ah = cx->A0h; e.ab_bstride = 0;
    e.B = 1; e.N = cx->N; e.R = cx->R; e.K = cx->K; e.lw = &cx->m->layers[0]; e.f16 = 1;
    return e;
}

// Builds the layer-0 message table of the complex: every intra-chain ordered pair (i, j) through the edge model once, features from
// the stored pose (any rigid placement of the ligand gives the same intra-chain geometry; k_edge_feat checks the bins per pose).
// Uses trajectory slot 0 of the workspace (ensure_workspace first) and overwrites ws.lig_cur's first pose.  Synchronises.
static EdgeArgs layer0_edge_args32(const dfm_complex *cx)      // fp32 engine: the complex's own fp32 A0 / Bm0
{
    EdgeArgs e;
    std::memset(&e, 0, sizeof(e));
    e.A = cx->A0; e.Bm = cx->Bm0; e.Bmb = cx->Bmb0; e.ab_bstride = 0;
    e.B = 1; e.N = cx->N; e.R = cx->R; e.K = cx->K; e.lw = &cx->m->layers[0];
    return e;
}

static int build_l0_table(dfm_complex *cx, float *build_ms, bool fp32 = false)
{
    Workspace &W = cx->ws;
    hipStream_t s = cx->stream;
    const size_t R = cx->R, L = cx->L, P = R * R + L * L;
    (fp32 ? cx->l0_valid32 : cx->l0_valid) = false;
    HIPCHK(hipStreamSynchronize(s));
    (fp32 ? cx->l0_pool32 : cx->l0_pool).release(true);
    cx->buf_gen++;
    uint2 *code0 = nullptr;
    if (fp32) {
        HIPCHK(cx->l0_pool32.alloc(&cx->l0_table32, P * H));
        HIPCHK(cx->l0_pool32.alloc(&cx->l0_code0_32, P));
        code0 = cx->l0_code0_32;
    } else {
        HIPCHK(cx->l0_pool.alloc(&cx->l0_table, ((P + 31) / 32 * 32) * H));
        HIPCHK(cx->l0_pool.alloc(&cx->l0_code0, P));
        code0 = cx->l0_code0;
    }
    DevPool tmp;
    tmp.bind(s);
    uint4 *rows = nullptr;
    HIPCHK(tmp.alloc(&rows, P));
    HIPCHK(hipEventRecord(cx->ev_total[0], s));
    HIPCHK(hipMemcpyAsync(W.lig_cur, cx->lig0, L * 9 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(launch_prep_pose(cx->rec_pos, W.lig_cur, 1, cx->R, cx->L, cx->m->hp.family == 1 ? 1 : 0, W.pos, W.ca4, W.cb4, s));
    HIPCHK(launch_l0_pairs(W.pos, W.ca4, W.cb4, cx->R, cx->L, cx->m->hp.mask_dist, code0, rows, s));
    if (fp32) HIPCHK(launch_edge_rows32(layer0_edge_args32(cx), rows, nullptr, (uint32_t)P, cx->l0_table32, s));
    else HIPCHK(launch_edge_rows(layer0_edge_args(cx), rows, nullptr, (uint32_t)P, cx->l0_table, s));
    HIPCHK(hipEventRecord(cx->ev_total[1], s));
    HIPCHK(hipStreamSynchronize(s));
    tmp.release(true);
    if (build_ms) HIPCHK(hipEventElapsedTime(build_ms, cx->ev_total[0], cx->ev_total[1]));
    (fp32 ? cx->l0_valid32 : cx->l0_valid) = true;
    return DFM_OK;
}

// Precision plan of the 16-bit MFMA engine (DFM_F_MFMA16).  Chosen on FOUR weight draws run through the reference - three
// seeds and one draw with every edge / node / coordinate MLP Linear times 3 (tests/golden/make_golden_draws.py,
// profiles/r03_exp_draw_knobs*.txt) - at SURVEY 8(d)'s gates (1e-2 on f / scores, 3e-2 on energy):
//   * per-edge contractions: fp16 operands in EVERY layer (same MFMA rate as bf16, 3 more mantissa bits).  With bf16 operands in
//     layers 0..4 (the r02 plan, now opt-in: DFM_F_BF16_OPS) the 3x-scaled draw reaches 1.5e-2 on f and 1.4e-2 on rot_score: the
//     rounding of the 256 x 256 weights to 8 bits is coherent over all edges and does not average out.  fp16 has no range cost
//     here: the gathered operands (Wb h_j, the lookup tables) are stored as fp16 in either plan.
//   * A_i = Wa h_i + b1 is read as fp16 in every layer (it joins Bm_j and the tables; fp32 A_i changes no worst case); the f16
//     engine (DFM_F_F16) keeps fp32.
//   * node-level GEMMs: three terms on split-bf16 operands (~1e-5).  The two-term fp16 form of r02 (weights as ONE fp16 tile) was
//     13 % faster per launch but its 2.4e-4 weight rounding is again coherent: 9.2e-3 on tr_score of the second family (seed 1);
//     it is gone.
extern "C" const char *dfm_config_string(void)
{
    static const std::string v = [] {
        std::string c = "mfma16: per-edge operands fp16 in every layer (DFM_F_BF16_OPS: bf16 in layers 0..depth-2), A_i fp16 "
                        "(DFM_F_F16: fp32), gathered Bm / tables fp16, node GEMMs ";
        c += "three-term split-bf16";
        c += ", fp32 accumulate / geometry / GraphNorm statistics / heads / SDE step";
        c += "; layer 0 through the per-complex message table in dfm_sample (DFM_F_NO_L0_TABLE: direct), on request in dfm_score (DFM_F_L0_TABLE)";
        c += "; build: TAB_MERGE=" + std::to_string((int)DFM_TAB_MERGE);
        std::string env;
        for (const char *k : {"DFM_EDGE_SPLIT", "DFM_GEMM_NARROW_MAXWG", "DFM_GEMM_QUARTER_MAXWG", "DFM_L0_TABLE", "DFM_GRAPH", "DFM_EDGE_F32_SCALAR", "DFM_GEMM_F32_SCALAR", "DFM_PAIR_HEAD_VALU", "DFM_PAIR_HEAD_M32", "DFM_ALLOC_CACHE", "DFM_ALLOC_CACHE_FRAC", "DFM_ALLOC_POISON", "DFM_ALLOC_GUARD", "DFM_LIB"}) {
            const char *e = getenv(k);
            if (e) env += std::string(env.empty() ? "" : " ") + k + "=" + e;
        }
        c += "; env: " + (env.empty() ? std::string("none") : env);
        return c;
    }();
    return v.c_str();
}

struct FwdOpts {
    bool bf16 = false, f16 = false, want_energy = false, profile = false;   // bf16: 16-bit MFMA engine; f16: ... with fp32 A_i
    bool bf16_ops = false;                // bf16 MFMA operands in every layer but the last (DFM_F_BF16_OPS)
    bool pose_prepared = false;           // ws.pos / ca4 / cb4 already hold this evaluation's pose (written by the previous step's k_heads)
    uint32_t *ctl = nullptr;              // replayed step graph: device {evaluation index, seed} block (ws.step_ctl) instead of by-value seed / stream id
    bool l0_table = false;                // layer 0 through the complex's message table (cx->l0_valid, workspace l0 buffers allocated)
    bool need_node_out = true;            // false: nobody reads the final node features (no energy / ires / debug tap) - the last layer then
                                          // computes the messages of the ligand nodes only (all the coordinate update reads) and no node model
    const int32_t *edges_dev = nullptr;   // [B][N][K] already on device (or nullptr = sample natively)
    int64_t edges_pitch = 0;              // elements between trajectories in edges_dev
    uint64_t seed = 0;
    float *h_first_out = nullptr;         // device [B][N][H] tap (dfm_score debug)
    // dfm_complex_selfcheck only: running maxima (float bits) [depth + 1][8] = per layer |h in|, |A|, |Bm|, |pre-activation 0|,
    // |pre-activation 2| of the fp32 pass; fp16 values found at the saturation value in A / Bm of a 16-bit pass
    uint32_t *range = nullptr;
    unsigned long long *sat = nullptr;
};

// largest |x| of a buffer -> *out (float bits; non-negative floats order like their bit patterns)
__global__ void k_absmax(const float *__restrict__ x, long long n, uint32_t *__restrict__ out)
{
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
// fp16 values at +-65504 (where every conversion of the 16-bit engine clamps) or beyond
__global__ void k_sat_count(const uint16_t *__restrict__ x, long long n, unsigned long long *__restrict__ out)
{
    unsigned c = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) c += (x[i] & 0x7fffu) >= 0x7bffu;
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}
static hipError_t launch_absmax(const float *x, long long n, uint32_t *out, hipStream_t s)
{
    const long long want = (n + 255) / 256;
    hipLaunchKernelGGL(k_absmax, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), token_lds(), s, x, n, out);
    return hipGetLastError();
}
static hipError_t launch_sat_count(const uint16_t *x, long long n, unsigned long long *out, hipStream_t s)
{
    const long long want = (n + 255) / 256;
    hipLaunchKernelGGL(k_sat_count, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), token_lds(), s, x, n, out);
    return hipGetLastError();
}

// One batched score evaluation of the poses in ws.lig_cur at times ws.t_dev; leaves f in ws.fvec, the
// node features in ws.h and (optionally) the energy partials.  Everything is enqueued on cx->stream.
static int enqueue_forward(dfm_complex *cx, int B, const FwdOpts &o)
{
    const dfm_model *m = cx->m;
    Workspace &W = cx->ws;
    hipStream_t s = cx->stream;
    const int N = cx->N, R = cx->R, L = cx->L, K = cx->K, depth = m->hp.depth;
    const uint32_t stream_id = cx->fwd_counter++;

    const bool pair_family = m->hp.family == 1;
    if (!o.pose_prepared) HIPCHK(launch_prep_pose(cx->rec_pos, W.lig_cur, B, R, L, pair_family ? 1 : 0, W.pos, W.ca4, W.cb4, s));
    if (o.edges_dev) {
        HIPCHK(hipMemcpy2DAsync(W.edges, (size_t)N * K * 4, o.edges_dev, (size_t)o.edges_pitch * 4, (size_t)N * K * 4, B,
                                hipMemcpyDeviceToDevice, s));
    } else {
        HIPCHK(launch_knn_sample(W.ca4, B, N, cx->knn, cx->nsamp, o.seed, stream_id, W.edges, o.ctl, s));
    }
    L0Classify cls;
    std::memset(&cls, 0, sizeof(cls));
    if (o.l0_table) { cls.code0 = o.bf16 ? cx->l0_code0 : cx->l0_code0_32; cls.src = W.l0_src; cls.rows = W.l0_rows; cls.counter = W.l0_counter; }
    HIPCHK(launch_edge_feat(W.pos, W.ca4, W.cb4, W.edges, B, N, R, K, m->hp.mask_dist, W.codes, W.radial, cls, o.ctl, s));
    // layer 0 reads the node embedding h0 [N][256] of the complex itself - identical for every trajectory - through a row period
    // (GemmArgs::a0_period / r_period) instead of a [B][N][256] copy made per evaluation (r01-r03: k_bcast_rows, 157 MB of writes at C3)
    const float *h = cx->h0;
    float *hn = W.h, *hspare = W.h2;
    const int M = B * N;
    const bool tile_tasks = o.bf16 && edge_msg_tile_tasks(B, N, K);      // (a ligand-only last layer decides for itself and zeroes agg if it must)
    for (int l = 0; l < depth; ++l) {
        const LayerDev &Lw = m->layers[l];
        const bool last = (l == depth - 1);
        const bool coord = last && !pair_family;       // EGNN_Net: update_coords = False in every layer
        // Score_Net reads the last layer's node outputs only in its energy / ires heads (score_net_mlsb.py:383-390); the force comes
        // from pos_out of the LIGAND nodes (:396-398), i.e. from the last layer's messages of ligand nodes alone
        const bool lig_only = coord && !o.need_node_out;
        EdgeArgs e;
        std::memset(&e, 0, sizeof(e));
        if (l == 0) { e.A = o.bf16 ? cx->A0s : cx->A0; e.Bm = cx->Bm0; e.Bmb = cx->Bmb0; e.ab_bstride = 0; }
        else { e.A = W.A; e.Bm = W.Bm; e.Bmb = W.Bmb; e.ab_bstride = (int64_t)N * H; }
        auto layer_f16 = [&](int ll) { return o.f16 || !o.bf16_ops || ll >= depth - 1; };
        auto layer_aw16 = [&](int) { return o.bf16 && !o.f16; };
        e.Ah = layer_aw16(l) ? (l == 0 ? cx->A0h : reinterpret_cast<const uint16_t *>(W.A)) : nullptr;
        e.edges = W.edges; e.codes = W.codes; e.radial = W.radial; e.ca4 = W.ca4;
        e.B = B; e.N = N; e.R = R; e.K = K; e.lw = &Lw; e.agg = W.agg; e.last = coord; e.fout = W.fvec; e.mbuf = W.mbuf;
        e.f16 = layer_f16(l) ? 1 : 0;
        e.lig_only = lig_only ? 1 : 0;
        e.agg_is_zero = (l > 0 && tile_tasks) ? 1 : 0;      // zeroed by the previous layer's node_mlp.3 GEMM
#if defined(DFM_EDGE_STAMP) || defined(DFM_EDGE_TRACE)
        e.stamp = (o.profile && l == 2) ? cx->stamp_dev : nullptr;
#else
        e.stamp = o.profile ? cx->stamp_dev : nullptr;      // every profiled message launch adds its clock stamps
#endif
        e.task_ctr = W.task_ctr;
        {   // selfcheck telemetry: what enters this layer
            const long long nrow = l == 0 ? N : M;      // layer 0's operands are per complex, not per trajectory
            if (o.range) {
                uint32_t *rg = o.range + l * 8;
                HIPCHK(launch_absmax(h, nrow * H, rg + 0, s));
                HIPCHK(launch_absmax(e.A, nrow * H, rg + 1, s));
                HIPCHK(launch_absmax(e.Bm, nrow * H, rg + 2, s));
                e.range = rg + 3;
            }
            if (o.sat) {
                if (e.Ah) HIPCHK(launch_sat_count(e.Ah, nrow * H, o.sat, s));
                HIPCHK(launch_sat_count(e.Bmb, nrow * H, o.sat, s));
            }
        }
        // the per-edge message kernel, bracketed by HIP events on this stream when profiling (dfm_get_profile)
        auto message_launch = [&](const EdgeArgs &ea) -> int {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (o.profile) {
                if (cx->ev_used + 2 > cx->ev.size()) {
                    hipEvent_t a, b2;
                    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b2));
                    cx->ev.push_back(a); cx->ev.push_back(b2);
                }
                e0 = cx->ev[cx->ev_used++]; e1 = cx->ev[cx->ev_used++];
                if (cx->ev_lig.size() < cx->ev_used / 2) cx->ev_lig.resize(cx->ev_used / 2);
                cx->ev_lig[cx->ev_used / 2 - 1] = ea.lig_only ? 1 : 0;
                HIPCHK(hipEventRecord(e0, s));
            }
            if (o.bf16) HIPCHK(launch_edge_bf16(ea, s)); else HIPCHK(launch_edge_f32(ea, s));
            if (o.profile) {
                HIPCHK(hipEventRecord(e1, s));
                cx->prof.edge_kernel_launches += 1;
                cx->prof.edge_lig_launches += ea.lig_only ? 1 : 0;
                cx->prof.edge_rows += (int64_t)ea.B * (ea.lig_only ? N - R : N) * K;
            }
            return DFM_OK;
        };
        {
            // (Running the last layer in trajectory chunks so that the stored messages stay in the Infinity Cache was measured and
            // dropped: 15 extra launch pairs per evaluation cost more - weight refills, partial rounds - than the round trip.)
            if (l == 0 && o.l0_table) {
                // layer 0 behind the message table: the edge model over the row list k_edge_feat left (inter-chain edges + bin
                // mismatches), then the K-row gather-sum in slot order from the table and that list's messages
                hipEvent_t t0 = nullptr, t1 = nullptr, t2 = nullptr;
                if (o.profile) {
                    while (cx->ev_l0_used + 3 > cx->ev_l0.size()) { hipEvent_t a; HIPCHK(hipEventCreate(&a)); cx->ev_l0.push_back(a); }
                    t0 = cx->ev_l0[cx->ev_l0_used++]; t1 = cx->ev_l0[cx->ev_l0_used++]; t2 = cx->ev_l0[cx->ev_l0_used++];
                    HIPCHK(hipEventRecord(t0, s));
                }
                if (o.bf16) HIPCHK(launch_edge_rows(e, W.l0_rows, W.l0_counter, (uint32_t)((size_t)B * N * K), W.l0_x, s));
                else HIPCHK(launch_edge_rows32(e, W.l0_rows, W.l0_counter, (uint32_t)((size_t)B * N * K), W.l0_x32, s));
                if (o.profile) HIPCHK(hipEventRecord(t1, s));
                if (o.bf16) HIPCHK(launch_l0_gather(cx->l0_table, W.l0_x, W.l0_src, W.agg, B, N, K, W.l0_counter, W.l0_miss_total, s));
                else HIPCHK(launch_l0_gather32(cx->l0_table32, W.l0_x32, W.l0_src, W.agg, B, N, K, W.l0_counter, W.l0_miss_total, s));
                if (o.profile) {
                    HIPCHK(hipEventRecord(t2, s));
                    cx->prof.l0_evals += 1; cx->prof.l0_edges += (int64_t)B * N * K;
                }
            } else {
                int rc2 = message_launch(e);
                if (rc2) return rc2;
            }
            if (coord && o.bf16) HIPCHK(launch_coord_bf16(e, s));
        }
        if (lig_only) break;      // the node model of the last layer feeds heads nobody asked for
        // node_model (egnn.py:106-116): u = Linear(cat[h, agg]); GraphNorm; SiLU; Linear; residual
        GemmArgs g;
        std::memset(&g, 0, sizeof(g));
        g.A0 = h; g.A1 = W.agg; g.lda = H; g.K = 2 * H; g.pro = 1; g.W = Lw.W3; g.ldw = 2 * H; g.bias = Lw.b3;
        g.M = M; g.Nout = H; g.C = W.u; g.ldc = H; g.a0_period = l == 0 ? N : 0;
        // 16-bit engines: the GEMM leaves per-tile column sums of u behind, so GraphNorm needs no extra pass over u
        const bool fused_stats = o.bf16 && gemm_rows_per_tile() == 64;
        if (fused_stats) { g.stat_part = W.gn_part; g.rows_per_graph = N; }
        if (o.bf16) HIPCHK(launch_gemm_split(g, Lw.W3_hi, Lw.W3_lo, s)); else HIPCHK(launch_gemm_f32(g, s));
        // fused_stats: the statistics are finished in the prologue of the next GEMM (no k_gn_finish launch)
        if (!fused_stats) HIPCHK(launch_gn_stats(W.u, B, N, Lw.gn_ms, W.gn_shift, W.gn_den, o.bf16 ? Lw.gn_w : nullptr, o.bf16 ? Lw.gn_b : nullptr, s));
        std::memset(&g, 0, sizeof(g));
        g.A0 = W.u; g.lda = H; g.K = H; g.pro = 2; g.gn_shift = W.gn_shift; g.gn_den = W.gn_den; g.gn_w = Lw.gn_w;
        g.gn_b = Lw.gn_b; g.rows_per_graph = N;
        if (fused_stats) { g.gn_part = W.gn_part; g.gn_ms = Lw.gn_ms; }
        // tile-task message launches add into a zero agg: its last reader (node_mlp.0) is done, this launch leaves it zeroed for the next layer
        if (o.bf16 && !last && tile_tasks) g.zbuf = W.agg; g.W = Lw.W4; g.ldw = H; g.bias = Lw.b4; g.M = M; g.Nout = H;
        g.epi = 1; g.R = h; g.C = hn; g.ldc = H; g.r_period = l == 0 ? N : 0;
        if (o.bf16) HIPCHK(launch_gemm_split(g, Lw.W4_hi, Lw.W4_lo, s)); else HIPCHK(launch_gemm_f32(g, s));
        { float *done = hn; hn = (l == 0) ? hspare : const_cast<float *>(h); h = done; }
        if (l == 0 && o.h_first_out)
            HIPCHK(hipMemcpyAsync(o.h_first_out, h, (size_t)M * H * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (!last) {   // next layer's per-node halves of edge_mlp.0: A = Wa h + b1, Bm = Wb h
            const LayerDev &Ln = m->layers[l + 1];
            std::memset(&g, 0, sizeof(g));
            g.A0 = h; g.lda = H; g.K = H; g.W = Ln.Wab; g.ldw = H; g.M = M; g.Nout = 2 * H;
            g.bias = cx->homomer ? (o.bf16 ? Ln.bias_ab_h_s : Ln.bias_ab_h) : (o.bf16 ? Ln.bias_ab_s : Ln.bias_ab);
            g.epi = 2; g.C = W.A; g.ldc = H; g.C2 = o.bf16 ? nullptr : W.Bm; g.C2b = W.Bmb;   // 16-bit engines gather the fp16 copy only
            if (layer_aw16(l + 1)) g.Cb = reinterpret_cast<uint16_t *>(W.A);     // ... and, bf16 operands, A as fp16 too (same buffer)
            if (o.bf16) HIPCHK(launch_gemm_split(g, Ln.Wab_hi, Ln.Wab_lo, s)); else HIPCHK(launch_gemm_f32(g, s));
        }
    }
    // keep the final node features in W.h (depth even).  The pair heads of family 1 read W.h on EVERY evaluation (run_head below),
    // the heads of family 0 only when somebody asked for the node outputs
    if (o.range && (o.need_node_out || pair_family)) HIPCHK(launch_absmax(h, (long long)M * H, o.range + depth * 8, s));
    if (h != W.h && (o.need_node_out || pair_family)) {
        HIPCHK(hipMemcpyAsync(W.h, h, (size_t)M * H * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (pair_family) {
        // egnn_net.py:430-470: pair heads on cat[h_r, h_l, D].  Per head: one GEMM projects every node through the stacked
        // halves of Linear(513 -> 256) (reusing the A / Bm buffers), then the elementwise pair kernel.
        static const bool pair_valu = [] { const char *e = getenv("DFM_PAIR_HEAD_VALU"); return e && atoi(e) != 0; }();      // A/B: r02-r03 kernel
        // 16-bit engines: k_pair_head_m (LayerNorm statistics from fp32 row moments + a dot product, 1-ulp hardware exp2 / rcp).
        // The fp32 engine - the baseline of dfm_complex_selfcheck and the fallback when a check fails - keeps the three-pass
        // LayerNorm, expf and IEEE division of k_pair_head<EXACT>: z = P + Q + w_d * D with D up to hundreds of Angstroms, so
        // E[z^2] - mean^2 loses digits to cancellation exactly when a trained checkpoint has a large mean / std ratio, which no
        // synthetic draw shows.  DFM_PAIR_HEAD_M32=1 opts the fp32 engine into the matrix-pipe kernel (+10 %).
        static const bool pair_m32 = [] { const char *e = getenv("DFM_PAIR_HEAD_M32"); return e && atoi(e) != 0; }();
        const bool pair_m = !pair_valu && (o.bf16 || pair_m32);
        auto run_head = [&](int q, int mode) -> int {
            const PairHeadDev &Ph = m->pair[q];
            GemmArgs g;
            std::memset(&g, 0, sizeof(g));
            g.A0 = W.h; g.lda = H; g.K = H; g.W = Ph.wab; g.ldw = H; g.M = M; g.Nout = 2 * H; g.epi = 2; g.C = W.A; g.ldc = H;
            g.C2 = W.Bm;
            if (o.bf16) HIPCHK(launch_gemm_split(g, Ph.wab_hi, Ph.wab_lo, s)); else HIPCHK(launch_gemm_f32(g, s));
            PairArgs a;
            std::memset(&a, 0, sizeof(a));
            a.P = W.A; a.Q = W.Bm; a.ca4 = W.ca4; a.B = B; a.R = R; a.L = L; a.w_d = Ph.w_d; a.ln_w = Ph.ln_w; a.ln_b = Ph.ln_b;
            a.w3 = Ph.w3; a.mode = mode; a.exact = o.bf16 ? 0 : 1; a.cut_off = m->hp.cut_off; a.fpart = W.fpart;
            a.spart = mode == 1 ? W.en_part : W.cpart; a.clash_part = W.clash_part;
            if (pair_m) {      // 16-bit engines: s(r, l) with the rank-4 part on the matrix pipe, then this head's reductions
                a.S = W.pair_s; a.Rp = (R + 31) / 32 * 32;
                HIPCHK(launch_pair_head_m(a, s));
                HIPCHK(launch_pair_finish_s(a, 4 * ((R + 63) / 64), m->hp.agg_mean ? 1.0f / (float)R : 1.0f, W.fvec, W.conf, s));
                return DFM_OK;
            }
            HIPCHK(launch_pair_head(a, s));
            return DFM_OK;
        };
        int rc = run_head(0, 0);
        if (rc) return rc;
        if (o.want_energy) {
            if ((rc = run_head(1, 1)) != DFM_OK) return rc;
            if ((rc = run_head(2, 2)) != DFM_OK) return rc;
        }
        if (!pair_m)
            HIPCHK(launch_pair_finish(W.fpart, B, R, L, m->hp.agg_mean ? 1.0f / (float)R : 1.0f, W.fvec, W.cpart,
                                      o.want_energy ? W.conf : nullptr, s));
    } else if (o.want_energy) {
        // to_energy.0 on cat[h_r, h_l] = Wa h_r + Wb h_l: project every node once (reuses A / Bm buffers)
        GemmArgs g;
        std::memset(&g, 0, sizeof(g));
        g.A0 = W.h; g.lda = H; g.K = H; g.W = m->heads.en_wa; g.ldw = 2 * H; g.M = M; g.Nout = H; g.C = W.A; g.ldc = H;
        HIPCHK(launch_gemm_f32(g, s));
        g.W = m->heads.en_wb; g.C = W.Bm;
        HIPCHK(launch_gemm_f32(g, s));
        HIPCHK(launch_energy_pairs(W.A, W.Bm, W.ca4, B, R, L, m->hp.cut_off, &m->heads, 1, W.en_part, W.clash_part, s));
    }
    return DFM_OK;
}

static void fill_head_args(dfm_complex *cx, int B, bool want_energy, HeadArgs *a)
{
    Workspace &W = cx->ws;
    std::memset(a, 0, sizeof(*a));
    a->fvec = W.fvec; a->ca4 = W.ca4; a->B = B; a->R = cx->R; a->L = cx->L; a->hid_base = W.hid_base; a->hid_bstride = 2 * HI; a->hw = &cx->m->heads;
    a->scores = W.scores; a->want_energy = want_energy; a->en_part = W.en_part; a->clash_part = W.clash_part;
    a->lig_cur = W.lig_cur; a->tr_update = W.tr_update; a->rot_update = W.rot_update;
    const dfm_hparams &hp = cx->m->hp;
    a->all_atoms = hp.family == 1;
    if (hp.family == 1) {
        a->n_part = 4 * ((cx->R + 63) / 64); a->en_mode = hp.agg_mean ? 1 : 2; a->pool_div = hp.agg_mean ? (float)cx->L : 1.0f;
    } else {
        a->n_part = cx->R; a->en_mode = 0; a->pool_div = (float)cx->L;
    }
}

static int finish_profile(dfm_complex *cx)
{
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, cx->ev_total[0], cx->ev_total[1]));
    cx->prof.total_ms = ms;
    for (size_t i = 0; i + 1 < cx->ev_used; i += 2) {
        HIPCHK(hipEventElapsedTime(&ms, cx->ev[i], cx->ev[i + 1]));
        cx->prof.edge_kernel_ms += ms;
        if (i / 2 < cx->ev_lig.size() && cx->ev_lig[i / 2]) cx->prof.edge_lig_ms += ms;
    }
    for (size_t i = 0; i + 2 < cx->ev_l0_used; i += 3) {
        HIPCHK(hipEventElapsedTime(&ms, cx->ev_l0[i], cx->ev_l0[i + 1]));
        cx->prof.l0_rows_ms += ms;
        HIPCHK(hipEventElapsedTime(&ms, cx->ev_l0[i + 1], cx->ev_l0[i + 2]));
        cx->prof.l0_gather_ms += ms;
    }
    if (cx->prof.l0_evals > 0 && cx->ws.l0_miss_total) {
        unsigned long long tot = 0;
        HIPCHK(hipMemcpyAsync(&tot, cx->ws.l0_miss_total, sizeof(tot), hipMemcpyDeviceToHost, cx->stream));
        HIPCHK(hipStreamSynchronize(cx->stream));
        cx->prof.l0_miss_rows = (int64_t)tot;
    }
#if !defined(DFM_EDGE_STAMP) && !defined(DFM_EDGE_TRACE)
    if (cx->stamp_dev && cx->ev_used) {      // the clock stamps of this call's message launches (k_edge_msg), then back to zero for the next call
        unsigned long long st[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(st, cx->stamp_dev, sizeof(st), hipMemcpyDeviceToHost, cx->stream));
        HIPCHK(hipMemsetAsync(cx->stamp_dev, 0, 2 * 8, cx->stream));
        HIPCHK(hipStreamSynchronize(cx->stream));
        cx->prof.edge_shader_cycles = (double)st[0];
        cx->prof.edge_ref_ticks = (double)st[1];
    }
#endif
#ifdef DFM_EDGE_TRACE      // diagnostic build: raw wave timelines of workgroup 0 -> $DFM_EDGE_TRACE_FILE (tools/edge_trace.py)
    if (cx->stamp_dev && getenv("DFM_EDGE_TRACE_FILE")) {
        static unsigned long long tr[48 + 8 * 130];      // [0..15]: s_memrealtime of every wave at its tiles 0 and 63; [48..]: the timelines
        HIPCHK(hipMemcpyAsync(tr, cx->stamp_dev, sizeof(tr), hipMemcpyDeviceToHost, cx->stream));
        HIPCHK(hipStreamSynchronize(cx->stream));
        if (FILE *f = fopen(getenv("DFM_EDGE_TRACE_FILE"), "wb")) { fwrite(tr, 1, sizeof(tr), f); fclose(f); }
    }
#endif
#ifdef DFM_EDGE_STAMP
    if (cx->stamp_dev) {
        unsigned long long st[48];
        HIPCHK(hipMemcpyAsync(st, cx->stamp_dev, sizeof(st), hipMemcpyDeviceToHost, cx->stream));
        HIPCHK(hipStreamSynchronize(cx->stream));
        for (int k = 0; k < 16; ++k) cx->prof.slot_cycles[k] = (double)st[32 + k];
        const double tiles = (double)cx->prof.edge_rows / (double)cx->prof.edge_kernel_launches / 32.0 / 2048.0;   // per wave
        for (int k = 0; k < 4; ++k) {
            double s2 = 0;
            for (int w = 0; w < 8; ++w) s2 += (double)st[w * 4 + k];
            cx->prof.phase_cycles[k] = s2 / 8.0 / (tiles > 0 ? tiles : 1.0);
        }
    }
#endif
    return DFM_OK;
}

extern "C" int dfm_get_profile(const dfm_complex *cx, dfm_profile *p)
{
    if (!cx || !p) return fail(DFM_E_INVALID, "NULL argument");
    *p = cx->prof;
    return DFM_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int dfm_score(dfm_complex *cx, int B, const float *lig_pos, const float *t, const int32_t *edges,
                         uint64_t seed, uint32_t flags, dfm_score_out *out)
{
    if (!cx || !lig_pos || !t || !out || !out->tr_score || !out->rot_score) return fail(DFM_E_INVALID, "NULL argument");
    if (B < 1) return fail(DFM_E_INVALID, "B must be >= 1");
    const bool f16 = flags & DFM_F_F16, bf16 = (flags & DFM_F_MFMA16) || f16, want_energy = flags & DFM_F_ENERGY;
    const bool want_ires = (flags & DFM_F_IRES) && out->ires;
    const bool want_dist = (flags & DFM_F_DIST) && out->dist_logits;
    if (want_dist && cx->m->hp.family != 1) return fail(DFM_E_INVALID, "DFM_F_DIST needs a family-1 model (EGNN_Net has to_dist, Score_Net does not)");
    DEVICE_SCOPE(cx->device);
    // layer 0 through the message table: on request only (dfm_score stays a pure function of its arguments and flags)
    const bool l0 = (flags & DFM_F_L0_TABLE) != 0;
    if (l0 && !(((bf16 && !f16 && !(flags & DFM_F_BF16_OPS)) || !bf16) && l0_eligible(cx, B)))
        return fail(DFM_E_INVALID, "DFM_F_L0_TABLE needs the fp32 or the DFM_F_MFMA16 engine (no DFM_F_F16 / DFM_F_BF16_OPS), depth >= 2 and a complex / batch within the table budgets");
    int rc = ensure_workspace(cx, B, bf16, l0);
    if (rc) return rc;
    Workspace &W = cx->ws;
    hipStream_t s = cx->stream;
    const size_t N = cx->N, L = cx->L, K = cx->K;
    cx->prof = dfm_profile{};
    cx->ev_used = 0; cx->ev_l0_used = 0;
    cx->fwd_counter = 0;   // RNG streams are a pure function of (seed, trajectory, evaluation index)
    if (l0) {
        if (!(bf16 ? cx->l0_valid : cx->l0_valid32)) {
            float bms = 0.f;
            if ((rc = build_l0_table(cx, &bms, !bf16)) != DFM_OK) return rc;
            cx->prof.l0_build_ms = bms;
        }
        HIPCHK(hipMemsetAsync(W.l0_miss_total, 0, sizeof(unsigned long long), s));
    }
    HIPCHK(hipMemcpyAsync(W.lig_cur, lig_pos, (size_t)B * L * 9 * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(W.t_dev, t, (size_t)B * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(launch_time_embed(W.t_dev, B, &cx->m->heads, W.hid_base, s));      // one entry per trajectory (fill_head_args: stride 2 * HI)
    DevPool tmp;   // per-call device buffers (injected edges, debug tap); released on every return path
    tmp.bind(cx->stream);
    int32_t *edges_dev = nullptr;
    float *h_first_dev = nullptr;
    if (edges) {
        HIPCHK(tmp.alloc(&edges_dev, (size_t)B * N * K));
        HIPCHK(hipMemcpyAsync(edges_dev, edges, (size_t)B * N * K * 4, hipMemcpyHostToDevice, s));
    }
    if (out->h_first) HIPCHK(tmp.alloc(&h_first_dev, (size_t)B * N * H));
    if (want_ires && !W.ir3) {      // kept with the workspace: a forward() loop does not pay three hipMalloc / hipFree pairs per call
        // (ir3 is the last of the three: a call that failed half-way allocates all of them again; the pool owns the orphans)
        W.ir1 = W.ir2 = nullptr;
        HIPCHK(W.pool.alloc(&W.ir1, (size_t)W.Bcap * N * 2 * H)); HIPCHK(W.pool.alloc(&W.ir2, (size_t)W.Bcap * N * 2 * H));
        HIPCHK(W.pool.alloc(&W.ir3, (size_t)W.Bcap * N));
    }
    float *ir1 = W.ir1, *ir2 = W.ir2, *ir3 = W.ir3;
    FwdOpts o;
    o.bf16 = bf16; o.f16 = f16; o.bf16_ops = bf16 && !f16 && (flags & DFM_F_BF16_OPS); o.want_energy = want_energy; o.profile = flags & DFM_F_PROFILE; o.edges_dev = edges_dev;
    o.edges_pitch = (int64_t)N * K; o.seed = seed; o.h_first_out = h_first_dev; o.l0_table = l0;
    // (h_first: a depth-1 model's first layer is its last - the tap needs that layer's node model too)
    o.need_node_out = want_energy || want_ires || want_dist || out->h_last != nullptr || out->h_first != nullptr;
    HIPCHK(hipEventRecord(cx->ev_total[0], s));
    rc = enqueue_forward(cx, B, o);
    if (rc == DFM_OK) {
        HeadArgs ha;
        fill_head_args(cx, B, want_energy, &ha);
        hipError_t e = launch_heads(ha, s);
        if (e != hipSuccess) rc = fail(DFM_E_HIP, hipGetErrorString(e));
    }
    if (rc == DFM_OK && want_ires) {
        // to_ires (score_net_mlsb.py:297-303,:383): Linear(256,512) SiLU Linear(512,512) SiLU Linear(512,1) on the final node
        // features; never read by the sampler, fp32 GEMMs in every engine
        const dfm_model *m = cx->m;
        GemmArgs g;
        std::memset(&g, 0, sizeof(g));
        g.A0 = W.h; g.lda = H; g.K = H; g.W = m->ir0_w; g.ldw = H; g.bias = m->ir0_b; g.M = (int)(B * N); g.Nout = 2 * H;
        g.C = ir1; g.ldc = 2 * H;
        hipError_t e = launch_gemm_f32(g, s);
        g.A0 = ir1; g.lda = 2 * H; g.K = 2 * H; g.pro = 3; g.W = m->ir2_w; g.ldw = 2 * H; g.bias = m->ir2_b; g.C = ir2;
        if (e == hipSuccess) e = launch_gemm_f32(g, s);
        g.A0 = ir2; g.W = m->ir4_w; g.bias = m->ir4_b; g.Nout = 1; g.C = ir3; g.ldc = 1;
        if (e == hipSuccess) e = launch_gemm_f32(g, s);
        if (e != hipSuccess) rc = fail(DFM_E_HIP, hipGetErrorString(e));
    }
    float *dist_dev = nullptr;
    if (rc == DFM_OK && want_dist) {
        // dist_logits (egnn_net.py:447): project every node through the stacked halves of to_dist.0 (reusing the A / Bm buffers),
        // then the pair kernel; fp32 in every engine
        const dfm_model *m = cx->m;
        hipError_t e = tmp.alloc(&dist_dev, (size_t)B * cx->R * L * 64);
        GemmArgs g;
        std::memset(&g, 0, sizeof(g));
        g.A0 = W.h; g.lda = H; g.K = H; g.W = m->dist.wab; g.ldw = H; g.M = (int)(B * N); g.Nout = 2 * H; g.epi = 2; g.C = W.A; g.ldc = H;
        g.C2 = W.Bm;
        if (e == hipSuccess) e = launch_gemm_f32(g, s);
        if (e == hipSuccess) e = launch_pair_dist(W.A, W.Bm, W.ca4, B, cx->R, (int)L, m->dist.w_d, m->dist.ln_w, m->dist.ln_b, m->dist.w3, dist_dev, s);
        if (e != hipSuccess) rc = fail(DFM_E_HIP, hipGetErrorString(e));
    }
    if (rc == DFM_OK) {
        std::vector<float> sc((size_t)B * 8);
        hipError_t e = hipEventRecord(cx->ev_total[1], s);
        if (e == hipSuccess) e = hipMemcpyAsync(sc.data(), W.scores, sc.size() * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && out->f) e = hipMemcpyAsync(out->f, W.fvec, (size_t)B * L * 3 * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && out->h_last) e = hipMemcpyAsync(out->h_last, W.h, (size_t)B * N * H * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && out->h_first) e = hipMemcpyAsync(out->h_first, h_first_dev, (size_t)B * N * H * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && out->edges) e = hipMemcpyAsync(out->edges, W.edges, (size_t)B * N * K * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && out->edge_codes) e = hipMemcpyAsync(out->edge_codes, W.codes, (size_t)B * N * K * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && want_ires) e = hipMemcpyAsync(out->ires, ir3, (size_t)B * N * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && want_dist) e = hipMemcpyAsync(out->dist_logits, dist_dev, (size_t)B * cx->R * L * 64 * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && out->confidence) {
            if (cx->m->hp.family == 1 && want_energy) e = hipMemcpyAsync(out->confidence, W.conf, (size_t)B * 4, hipMemcpyDeviceToHost, s);
            else std::memset(out->confidence, 0, (size_t)B * 4);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = fail(DFM_E_HIP, hipGetErrorString(e));
        else {
            for (int b = 0; b < B; ++b) {
                for (int k = 0; k < 3; ++k) { out->tr_score[b * 3 + k] = sc[b * 8 + k]; out->rot_score[b * 3 + k] = sc[b * 8 + 3 + k]; }
                if (out->energy) out->energy[b] = sc[b * 8 + 6];
                if (out->num_clashes) out->num_clashes[b] = (int32_t)sc[b * 8 + 7];
            }
            if (o.profile) rc = finish_profile(cx);
        }
    }
    if (rc != DFM_OK) (void)hipStreamSynchronize(s);   // nothing may still read the per-call buffers when they are released
    return rc;
}

// ------------------------------------------------------------------------------------------------
extern "C" int dfm_sample(dfm_complex *cx, int B, int num_steps, float eps, float tr_noise_scale, float rot_noise_scale,
                          uint32_t flags, uint64_t seed, const dfm_inject *inj, dfm_traj_out *out)
{
    if (!cx || !out) return fail(DFM_E_INVALID, "NULL argument");
    if (B < 1 || num_steps < 2) return fail(DFM_E_INVALID, "need B >= 1 and num_steps >= 2");
    if (num_steps > (1 << 24)) return fail(DFM_E_INVALID, "num_steps above 2^24");
    const bool f16 = flags & DFM_F_F16, bf16 = (flags & DFM_F_MFMA16) || f16;
    DEVICE_SCOPE(cx->device);
    // layer 0 through the complex's message table whenever the shipped 16-bit plan (or the fp32 engine) runs and the complex is
    // eligible - a property of the complex alone (l0_eligible), so a trajectory's bits do not depend on the batch it is sampled in
    const bool l0 = ((bf16 && !f16 && !(flags & DFM_F_BF16_OPS)) || !bf16) && !(flags & DFM_F_NO_L0_TABLE) && l0_eligible(cx, B);
    int rc = ensure_workspace(cx, B, bf16, l0);
    if (rc) return rc;
    if ((rc = ensure_time_grid(cx, (size_t)(B > num_steps ? B : num_steps))) != DFM_OK) return rc;
    Workspace &W = cx->ws;
    hipStream_t s = cx->stream;
    const dfm_hparams &hp = cx->m->hp;
    const size_t N = cx->N, L = cx->L, K = cx->K, S = num_steps;
    cx->prof = dfm_profile{};
    cx->ev_used = 0; cx->ev_l0_used = 0;
    cx->fwd_counter = 0;   // evaluation i of this call draws its graph from Philox stream i
    if (l0) {
        if (!(bf16 ? cx->l0_valid : cx->l0_valid32)) {
            float bms = 0.f;
            if ((rc = build_l0_table(cx, &bms, !bf16)) != DFM_OK) return rc;
            cx->prof.l0_build_ms = bms;
        }
        HIPCHK(hipMemsetAsync(W.l0_miss_total, 0, sizeof(unsigned long long), s));
    }

    // time grid: torch.linspace(1, eps, num_steps) in float32; dt = t[0] - t[1]   (inference_base.py:404-405)
    std::vector<float> ts(S);
    {
        const float step = (eps - 1.0f) / (float)(num_steps - 1);
        for (int i = 0; i < num_steps; ++i)
            ts[i] = i < num_steps / 2 ? 1.0f + step * (float)i : eps - step * (float)(num_steps - 1 - i);
    }
    const float dt = ts[0] - ts[1];
    std::vector<double> gr(S), gt(S);
    for (int i = 0; i < num_steps; ++i) {
        rc = dfm_diffusion_coef(&hp, 1, (double)ts[i], &gr[i], nullptr);   // ValueError parity: t outside [0,1]
        if (rc) return rc;
        dfm_diffusion_coef(&hp, 0, (double)ts[i], &gt[i], nullptr);
    }

    DevPool tmp;   // per-call device buffers (injections, traces)
    tmp.bind(cx->stream);
    float *R0_d = nullptr, *trd_d = nullptr, *zr_d = nullptr, *zt_d = nullptr, *tp_d = nullptr, *tsc_d = nullptr,
          *ip_d = nullptr;
    int32_t *ed_d = nullptr;
    if (inj) {
        if (inj->R0) HIPCHK(tmp.upload(&R0_d, inj->R0, (size_t)B * 9));
        if (inj->tr_draw) HIPCHK(tmp.upload(&trd_d, inj->tr_draw, (size_t)B * 3));
        if (inj->z_rot) HIPCHK(tmp.upload(&zr_d, inj->z_rot, (size_t)B * S * 3));
        if (inj->z_tr) HIPCHK(tmp.upload(&zt_d, inj->z_tr, (size_t)B * S * 3));
        if (inj->edges) HIPCHK(tmp.upload(&ed_d, inj->edges, (size_t)B * (S + 1) * N * K));
    }
    if (out->trace_pose) HIPCHK(tmp.alloc(&tp_d, (size_t)B * S * L * 9));
    if (out->trace_scores) HIPCHK(tmp.alloc(&tsc_d, (size_t)B * (S + 1) * 8));

    // every trajectory of a step shares its time: the time-dependent half of the score-scale MLPs for the whole grid, once per call
    HIPCHK(hipMemcpyAsync(W.t_dev, ts.data(), S * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(launch_time_embed(W.t_dev, num_steps, &cx->m->heads, W.hid_base, s));
    HIPCHK(hipEventRecord(cx->ev_total[0], s));
    HIPCHK(launch_init_pose(cx->rec_pos, cx->lig0, B, cx->R, cx->L, hp.family == 1, R0_d, trd_d, seed, W.lig_cur, W.tr_update,
                            W.rot_update, s));
    if (out->init_pose) {
        HIPCHK(tmp.alloc(&ip_d, (size_t)B * L * 9));
        HIPCHK(hipMemcpyAsync(ip_d, W.lig_cur, (size_t)B * L * 9 * 4, hipMemcpyDeviceToDevice, s));
    }
    FwdOpts o;
    o.bf16 = bf16; o.f16 = f16; o.bf16_ops = bf16 && !f16 && (flags & DFM_F_BF16_OPS); o.profile = flags & DFM_F_PROFILE; o.seed = seed; o.edges_pitch = (int64_t)(S + 1) * N * K;
    o.l0_table = l0;
    const bool step_energy = (flags & DFM_F_STEP_ENERGY) != 0;
    // the by-value arguments of step i's k_heads launch
    auto step_head_args = [&](int i, HeadArgs &ha) {
        const bool is_last = (i == num_steps - 1);
        fill_head_args(cx, B, step_energy, &ha);
        ha.hid_base = W.hid_base + (size_t)i * 2 * HI; ha.hid_bstride = 0;      // this step's time for the whole batch
        ha.do_update = 1;
        ha.g2_r = (float)(gr[i] * gr[i]); ha.g_r = (float)gr[i]; ha.hg2_r = (float)(0.5 * (gr[i] * gr[i]));
        ha.g2_t = (float)(gt[i] * gt[i]); ha.g_t = (float)gt[i]; ha.hg2_t = (float)(0.5 * (gt[i] * gt[i]));
        ha.dt = dt; ha.sqrt_dt = std::sqrt(dt);
        if (flags & DFM_F_NOISE_ANNEALING) { ha.tr_noise = ts[i]; ha.rot_noise = ts[i]; }       // inference_base.py:428-430
        else { ha.tr_noise = is_last ? 0.0f : tr_noise_scale; ha.rot_noise = is_last ? 0.0f : rot_noise_scale; }
        ha.ode = (flags & DFM_F_ODE) ? 1 : 0;
        ha.z_rot = zr_d ? zr_d + (size_t)i * 3 : nullptr; ha.z_tr = zt_d ? zt_d + (size_t)i * 3 : nullptr;
        ha.z_bstride = (int64_t)S * 3; ha.seed = seed; ha.step = (uint32_t)i;
        if (tp_d) { ha.trace_pose = tp_d + (size_t)i * L * 9; ha.trace_bstride = (int64_t)S * L * 9; }
        if (tsc_d) { ha.trace_scores = tsc_d + (size_t)i * 8; ha.trace_s_bstride = (int64_t)(S + 1) * 8; }
    };
    // Replayed step graph (DFM_F_GRAPH, or DFM_GRAPH=1 in the environment): nothing injected, traced or timed per launch.  Same
    // kernels with the same arguments as the plain path - bitwise the same trajectories (tests/test_gpu_graph.py).
    static const bool graph_env_on = [] { const char *e = getenv("DFM_GRAPH"); return e && atoi(e) != 0; }();
    const bool use_graph = (graph_env_on || (flags & DFM_F_GRAPH)) && !(flags & (DFM_F_PROFILE | DFM_F_STEP_ENERGY)) && !inj && !tp_d && !tsc_d;
    // the pose a step produces is prepared for the next evaluation by that step's k_heads (one dependent launch less per step) unless
    // something still moves it afterwards (clash force) or the step is a captured graph (whose first launch is the preparation)
    const bool fuse_prep = !use_graph && !(flags & DFM_F_CLASH_FORCE);
    if (use_graph) {
        std::vector<StepParams> sp(S);
        for (int i = 0; i < num_steps; ++i) {
            HeadArgs ha;
            step_head_args(i, ha);
            sp[i] = StepParams{ha.g2_r, ha.g_r, ha.hg2_r, ha.g2_t, ha.g_t, ha.hg2_t, ha.dt, ha.sqrt_dt, ha.rot_noise, ha.tr_noise, ha.step, 0u};
        }
        const uint32_t ctl_h[4] = {0u, (uint32_t)seed, (uint32_t)(seed >> 32), 0u};
        HIPCHK(hipMemcpyAsync(W.step_params, sp.data(), S * sizeof(StepParams), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(W.step_ctl, ctl_h, sizeof(ctl_h), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));      // (pageable sources: the copies have read them)
        const uint64_t key = ((uint64_t)(uint32_t)B << 32) | (uint64_t)(flags & (DFM_F_MFMA16 | DFM_F_F16 | DFM_F_BF16_OPS | DFM_F_CLASH_FORCE | DFM_F_ODE | DFM_F_NO_L0_TABLE))
                             | (l0 ? 1ull << 31 : 0ull);
        if (!cx->step_exec || cx->step_key != key || cx->step_gen != cx->buf_gen) {
            if (cx->step_exec) { (void)hipGraphExecDestroy(cx->step_exec); cx->step_exec = nullptr; }
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            o.ctl = W.step_ctl; o.edges_dev = nullptr; o.want_energy = false; o.need_node_out = false;
            int crc = enqueue_forward(cx, B, o);
            if (crc == DFM_OK) {
                HeadArgs ha;
                step_head_args(0, ha);
                ha.hid_base = W.hid_base; ha.step_params = W.step_params; ha.ctl = W.step_ctl;
                hipError_t e = launch_heads(ha, s);
                if (e == hipSuccess && (flags & DFM_F_CLASH_FORCE)) e = launch_clash_force(cx->rec_pos, B, cx->R, cx->L, W.lig_cur, W.tr_update, s);
                if (e != hipSuccess) crc = fail(DFM_E_HIP, hipGetErrorString(e));
            }
            const std::string keep = g_err;
            hipError_t ee = hipStreamEndCapture(s, &graph);      // always: a stream left in capture mode is unusable
            o.ctl = nullptr;
            if (crc != DFM_OK) { if (graph) (void)hipGraphDestroy(graph); g_err = keep; return crc; }
            if (ee != hipSuccess) return fail(DFM_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ee));
            ee = hipGraphInstantiate(&cx->step_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ee != hipSuccess) { cx->step_exec = nullptr; return fail(DFM_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ee)); }
            cx->step_key = key; cx->step_gen = cx->buf_gen;
        }
        for (int i = 0; i < num_steps; ++i) HIPCHK(hipGraphLaunch(cx->step_exec, s));
        cx->fwd_counter = (uint32_t)num_steps;      // the final evaluation below draws its graph from Philox stream num_steps
    } else
    for (int i = 0; i < num_steps; ++i) {
        o.edges_dev = ed_d ? ed_d + (size_t)i * N * K : nullptr;
        o.want_energy = step_energy; o.need_node_out = step_energy;
        o.pose_prepared = fuse_prep && i > 0;
        rc = enqueue_forward(cx, B, o);
        if (rc) return rc;
        HeadArgs ha;
        step_head_args(i, ha);
        if (fuse_prep) { ha.prep_next = 1; ha.rec_pos = cx->rec_pos; ha.prep_pos = W.pos; ha.prep_ca4 = W.ca4; ha.prep_cb4 = W.cb4; }
        HIPCHK(launch_heads(ha, s));
        if (flags & DFM_F_CLASH_FORCE) {   // inference_base.py:458-461
            HIPCHK(launch_clash_force(cx->rec_pos, B, cx->R, cx->L, W.lig_cur, W.tr_update, s));
            if (tp_d) HIPCHK(hipMemcpy2DAsync(tp_d + (size_t)i * L * 9, S * L * 9 * 4, W.lig_cur, L * 9 * 4, L * 9 * 4, B,
                                              hipMemcpyDeviceToDevice, s));
        }
    }
    // final evaluation of the last pose, with the energy head (inference_base.py:463-466); same t as the last step
    o.edges_dev = ed_d ? ed_d + (size_t)num_steps * N * K : nullptr;
    o.want_energy = true; o.need_node_out = true; o.pose_prepared = fuse_prep;
    rc = enqueue_forward(cx, B, o);
    if (rc) return rc;
    {
        HeadArgs ha;
        fill_head_args(cx, B, true, &ha);
        ha.hid_base = W.hid_base + (size_t)(num_steps - 1) * 2 * HI; ha.hid_bstride = 0;      // same t as the last step
        if (tsc_d) { ha.trace_scores = tsc_d + (size_t)num_steps * 8; ha.trace_s_bstride = (int64_t)(S + 1) * 8; }
        HIPCHK(launch_heads(ha, s));
    }
    HIPCHK(hipEventRecord(cx->ev_total[1], s));
    std::vector<float> sc((size_t)B * 8);
    HIPCHK(hipMemcpyAsync(sc.data(), W.scores, sc.size() * 4, hipMemcpyDeviceToHost, s));
    if (out->lig_pos) HIPCHK(hipMemcpyAsync(out->lig_pos, W.lig_cur, (size_t)B * L * 9 * 4, hipMemcpyDeviceToHost, s));
    if (out->rot_update) HIPCHK(hipMemcpyAsync(out->rot_update, W.rot_update, (size_t)B * 3 * 4, hipMemcpyDeviceToHost, s));
    if (out->tr_update) HIPCHK(hipMemcpyAsync(out->tr_update, W.tr_update, (size_t)B * 3 * 4, hipMemcpyDeviceToHost, s));
    if (tp_d) HIPCHK(hipMemcpyAsync(out->trace_pose, tp_d, (size_t)B * S * L * 9 * 4, hipMemcpyDeviceToHost, s));
    if (tsc_d) HIPCHK(hipMemcpyAsync(out->trace_scores, tsc_d, (size_t)B * (S + 1) * 8 * 4, hipMemcpyDeviceToHost, s));
    if (ip_d) HIPCHK(hipMemcpyAsync(out->init_pose, ip_d, (size_t)B * L * 9 * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int b = 0; b < B; ++b) {
        if (out->energy) out->energy[b] = sc[b * 8 + 6];
        if (out->num_clashes) out->num_clashes[b] = (int32_t)sc[b * 8 + 7];
        if (out->final_scores) std::memcpy(out->final_scores + (size_t)b * 6, &sc[(size_t)b * 8], 6 * sizeof(float));
    }
    if (o.profile) return finish_profile(cx);
    return DFM_OK;
}

// ------------------------------------------------------------------------------------------------
// Runtime parity evidence for weights the build has never seen (include/dfmdock_amd.h: dfm_selfcheck_out).
extern "C" int dfm_complex_selfcheck(dfm_complex *cx, int n_eval, const float *t_in, uint64_t seed, uint32_t flags, dfm_selfcheck_out *out)
{
    if (!cx || !out) return fail(DFM_E_INVALID, "NULL argument");
    if (n_eval < 1 || n_eval > 16) return fail(DFM_E_INVALID, "n_eval must be in 1..16");
    DEVICE_SCOPE(cx->device);
    const int B = n_eval;
    // the 16-bit pass runs the engine as dfm_sample runs it: layer 0 through the message table where the complex is eligible
    const bool l0 = !(flags & (DFM_F_F16 | DFM_F_BF16_OPS | DFM_F_NO_L0_TABLE)) && l0_eligible(cx, B);
    int rc = ensure_workspace(cx, B, true, l0);
    if (rc) return rc;
    if (l0 && !cx->l0_valid && (rc = build_l0_table(cx, nullptr)) != DFM_OK) return rc;
    Workspace &W = cx->ws;
    hipStream_t s = cx->stream;
    const dfm_model *m = cx->m;
    const int depth = m->hp.depth;
    const size_t N = cx->N, L = cx->L, K = cx->K;
    std::vector<float> t(B);
    for (int b = 0; b < B; ++b) {
        t[b] = t_in ? t_in[b] : (B == 1 ? 0.5f : 1.0f + (0.001f - 1.0f) * (float)b / (float)(B - 1));
        if (!(t[b] >= 0.f && t[b] <= 1.f)) return fail(DFM_E_INVALID, "Invalid t (need 0 <= t <= 1)");
    }
    cx->prof = dfm_profile{};
    cx->ev_used = 0; cx->ev_l0_used = 0;
    cx->fwd_counter = 0;
    DevPool tmp;
    tmp.bind(s);
    uint32_t *range_d = nullptr;
    unsigned long long *sat_d = nullptr;
    int32_t *edges_d = nullptr;
    HIPCHK(tmp.alloc(&range_d, (size_t)(depth + 1) * 8)); HIPCHK(tmp.alloc(&sat_d, 1)); HIPCHK(tmp.alloc(&edges_d, (size_t)B * N * K));
    HIPCHK(hipMemsetAsync(range_d, 0, (size_t)(depth + 1) * 8 * 4, s)); HIPCHK(hipMemsetAsync(sat_d, 0, 8, s));
    for (int b = 0; b < B; ++b)
        HIPCHK(hipMemcpyAsync(W.lig_cur + (size_t)b * L * 9, cx->lig0, L * 9 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(W.t_dev, t.data(), (size_t)B * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(launch_time_embed(W.t_dev, B, &cx->m->heads, W.hid_base, s));

    struct Res { std::vector<float> sc, f; };
    auto run = [&](bool mfma, Res &r) -> int {
        FwdOpts o;
        o.bf16 = mfma; o.f16 = mfma && (flags & DFM_F_F16); o.bf16_ops = mfma && !o.f16 && (flags & DFM_F_BF16_OPS);
        o.want_energy = true; o.need_node_out = true; o.seed = seed;
        if (mfma) {
            o.edges_dev = edges_d; o.edges_pitch = (int64_t)N * K; o.sat = sat_d; o.l0_table = l0;
            if (l0)      // the table's own entries (S * gate * m as fp16) are clamped like every other 16-bit store: count them too
                HIPCHK(launch_sat_count(cx->l0_table, (long long)((size_t)cx->R * cx->R + (size_t)cx->L * cx->L) * H, sat_d, s));
        } else o.range = range_d;
        int rc2 = enqueue_forward(cx, B, o);
        if (rc2) return rc2;
        HeadArgs ha;
        fill_head_args(cx, B, true, &ha);
        HIPCHK(launch_heads(ha, s));
        r.sc.resize((size_t)B * 8); r.f.resize((size_t)B * L * 3);
        HIPCHK(hipMemcpyAsync(r.sc.data(), W.scores, r.sc.size() * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(r.f.data(), W.fvec, r.f.size() * 4, hipMemcpyDeviceToHost, s));
        if (!mfma) HIPCHK(hipMemcpyAsync(edges_d, W.edges, (size_t)B * N * K * 4, hipMemcpyDeviceToDevice, s));      // the graphs the fp32 pass drew
        HIPCHK(hipStreamSynchronize(s));
        return DFM_OK;
    };
    Res r32, r16;
    std::vector<float4> ca(N);      // centred CA of the (one) pose: r of the torque pooling
    if ((rc = run(false, r32)) != DFM_OK) { (void)hipStreamSynchronize(s); return rc; }
    HIPCHK(hipMemcpyAsync(ca.data(), W.ca4, N * sizeof(float4), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if ((rc = run(true, r16)) != DFM_OK) { (void)hipStreamSynchronize(s); return rc; }
    std::vector<uint32_t> rg((size_t)(depth + 1) * 8);
    unsigned long long sat = 0;
    HIPCHK(hipMemcpyAsync(rg.data(), range_d, rg.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&sat, sat_d, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));

    dfm_selfcheck_out &o = *out;
    std::memset(&o, 0, sizeof(o));
    o.n_eval = B; o.depth = depth;
    o.gate_f = 1e-2f; o.gate_score = 1e-2f; o.gate_energy = 3e-2f; o.limit = 6.0e4f;
    o.cancel_ratio[0] = o.cancel_ratio[1] = 1e30f;
    auto dev_of = [](const float *a, const float *ref, size_t n) {      // L-inf over L-inf; NaN / inf anywhere -> NaN (never "ok")
        double d = 0, mx = 0;
        for (size_t i = 0; i < n; ++i) {
            if (!std::isfinite(a[i]) || !std::isfinite(ref[i])) return std::nanf("");
            d = std::fmax(d, std::fabs((double)a[i] - (double)ref[i])); mx = std::fmax(mx, std::fabs((double)ref[i]));
        }
        return (float)(d / std::fmax(mx, 1e-30));
    };
    auto worse = [](float cur, float v) { return (v != v || cur != cur) ? std::nanf("") : std::fmax(cur, v); };
    const size_t R = cx->R;
    for (int b = 0; b < B; ++b) {
        const float *f32 = &r32.f[(size_t)b * L * 3], *f16 = &r16.f[(size_t)b * L * 3];
        const float df = dev_of(f16, f32, L * 3);
        o.dev_f = worse(o.dev_f, df);
        o.dev_tr_score = worse(o.dev_tr_score, dev_of(&r16.sc[(size_t)b * 8], &r32.sc[(size_t)b * 8], 3));
        o.dev_rot_score = worse(o.dev_rot_score, dev_of(&r16.sc[(size_t)b * 8 + 3], &r32.sc[(size_t)b * 8 + 3], 3));
        const double e32 = r32.sc[(size_t)b * 8 + 6], e16 = r16.sc[(size_t)b * 8 + 6];
        o.dev_energy = worse(o.dev_energy, (std::isfinite(e32) && std::isfinite(e16)) ? (float)(std::fabs(e16 - e32) / std::fmax(std::fabs(e32), 0.1)) : std::nanf(""));
        // pooled force mean_l f and torque mean_l (r_l x f_l) (score_net_mlsb.py:396-405): cancellation ratios and the bounds they give
        double mf[3] = {0, 0, 0}, mt[3] = {0, 0, 0}, af = 0, at = 0, fmax = 0, rmean = 0;
        for (size_t l = 0; l < L; ++l) {
            const float *v = f32 + l * 3;
            const float4 r4 = ca[R + l];
            const double r[3] = {r4.x, r4.y, r4.z};
            const double c[3] = {r[1] * v[2] - r[2] * v[1], r[2] * v[0] - r[0] * v[2], r[0] * v[1] - r[1] * v[0]};
            for (int k = 0; k < 3; ++k) { mf[k] += v[k]; mt[k] += c[k]; fmax = std::fmax(fmax, std::fabs((double)v[k])); }
            af += std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]);
            at += std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
            rmean += std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        }
        const double nf = std::sqrt(mf[0] * mf[0] + mf[1] * mf[1] + mf[2] * mf[2]) / (double)L;      // |mean f|
        const double nt = std::sqrt(mt[0] * mt[0] + mt[1] * mt[1] + mt[2] * mt[2]) / (double)L;      // |mean r x f|
        af /= (double)L; at /= (double)L; rmean /= (double)L;
        o.cancel_ratio[0] = std::fmin(o.cancel_ratio[0], (float)(nf / std::fmax(af, 1e-30)));
        o.cancel_ratio[1] = std::fmin(o.cancel_ratio[1], (float)(nt / std::fmax(at, 1e-30)));
        const double dabs = std::sqrt(3.0) * (double)df * fmax;      // |d f_l| <= sqrt(3) dev_f max|f| for every l
        o.score_bound[0] = worse(o.score_bound[0], (float)(dabs / std::fmax(nf, 1e-30)));
        o.score_bound[1] = worse(o.score_bound[1], (float)(dabs * rmean / std::fmax(nt, 1e-30)));
    }
    const float LOG2E = -SILU_S;
    auto val = [&](int l, int k) { return __builtin_bit_cast(float, rg[(size_t)l * 8 + k]); };
    float worst = 0.f;
    for (int l = 0; l <= depth; ++l) o.max_h[l] = val(l, 0);
    for (int l = 0; l < depth; ++l) {
        o.max_A[l] = LOG2E * val(l, 1); o.max_Bm[l] = LOG2E * val(l, 2);
        o.max_tab[l] = std::fmax(m->tab_max[l][0], m->tab_max[l][1]);
        o.max_sum16[l] = o.max_Bm[l] + m->tab_max[l][0] + m->tab_max[l][1];
        o.max_pre[l] = LOG2E * val(l, 3); o.max_acc[l] = LOG2E * val(l, 4);
        for (float v : {o.max_A[l], o.max_Bm[l], o.max_tab[l], o.max_sum16[l], o.max_pre[l], o.max_acc[l]}) worst = (v != v) ? INFINITY : std::fmax(worst, v);
    }
    o.headroom = worst > 0.f ? o.limit / worst : INFINITY;
    o.saturated = (int64_t)sat;
    o.range_ok = (worst < o.limit && sat == 0) ? 1 : 0;
    const float gate_tr = std::fmax(o.gate_score, 2.0f * o.score_bound[0]), gate_rot = std::fmax(o.gate_score, 2.0f * o.score_bound[1]);
    o.dev_ok = (o.dev_f <= o.gate_f && o.dev_energy <= o.gate_energy && o.dev_tr_score <= gate_tr && o.dev_rot_score <= gate_rot) ? 1 : 0;
    o.ok = o.range_ok && o.dev_ok;
    return DFM_OK;
}

// Engine: the whole tdnn (model/tdnn.py:33-191) + entire_network (model/trainer.py:168-188) +
// loss (model/loss.py) + regulariser/optimiser (model/trainer.py:332-436) graph as a fixed
// sequence of kernel launches on one HIP stream.  This is the native counterpart of what the
// TF1 runtime does under sess.run(train_op); the Python Trainer only feeds pointers.
//
// HBM layout (all fp32, channel axis contiguous):
//   variables  : caller-owned flat buffer, TF variable order, trainable first then BN moving
//                statistics; every variable starts on a 16-byte boundary.  Gradients mirror the
//                trainable section, so backward "stages" finish contiguous tail slices of the
//                gradient buffer and the host can all-reduce them while earlier layers still run.
//   activations: per frame layer z_l (pre-BN) and a_l (post BN+ReLU), [chunks*frames_l][C_l].
//   backward   : one da buffer and one dz buffer, ping-ponged down the stack; dz is stored with
//                k-1 zero frames around each chunk so the data gradient is the SAME spliced-view
//                GEMM as the forward pass (xv_gemm.hip).
//   weights    : kernel-layout copies (transposed / tap-flipped / channel-padded) rebuilt once
//                per optimiser step.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "xv_common.h"

namespace {

struct Var {
    std::string name;
    int32_t shape[4];
    int32_t rank;
    size_t offset;   // floats, into the variables buffer
    size_t count;    // floats
    bool trainable;
};

struct Affine {   // one conv/dense layer (+ optional BN, ReLU)
    std::string prefix;    // "tdnn1"
    std::string kind;      // "conv" | "dense"
    int k, c_in, c_pad, c_out;
    bool has_bn, has_relu, fused_bn;
    int v_kernel, v_bias, v_gamma, v_beta, v_mmean, v_mvar;
    float *wt, *wf;                       // kernel-layout weights (wf only when k > 1)
    float *z, *a;                         // activations
    float *bn_part, *mean, *invstd, *scale, *shift;
    // split precision (f16x3): fp16 planes of the kernel-layout weights and of this layer's BN+ReLU output
    unsigned short *wth = nullptr, *wfh = nullptr, *ah = nullptr;
    size_t wth_stride = 0, wfh_stride = 0;    // plane strides (elements)
    int o_ld = 0;                             // plane pitch of c_out (multiple of 8)
    int ldz = 0;                              // floats per row of z and of this layer's dz (= c_out except the pooled layer: rows on the 128-byte grid)
    float *zmin = nullptr, *zmax = nullptr;
    int rows;                             // rows of the most recent forward
    std::string scope;                    // variable scope under "tdnn/" ("" or "attention/att_key0/")
    int in_layer = -1;                    // index of the layer whose output this one reads (-1: the features / the pooled vector)
    int act = 0;                          // 3: tanh on the affine output (att_key_network_type 3), no BN
    int wslot = 0, aslot = 0;             // amax slots of the weights / of the BN+ReLU output planes
    int v_alpha = -1;                     // prelu: the layer's "<prefix>_relu/alpha" variable (network_relu_type, common.py:35-39)
};

}  // namespace

constexpr int XV_Z_SLOTS = 16;

struct xv_engine {
    xv_config cfg;
    std::vector<Var> vars;
    size_t n_train = 0, n_all = 0, n_opt = 0;
    float *V = nullptr, *G = nullptr, *S = nullptr;   // bound buffers
    // frame-level layers tdnn1..tdnnF (F = 5 in the reference, tdnn.py:35-127; any table of (context, width) in the extended
    // form), then the two segment-level layers tdnn(F+1), tdnn(F+2), then the attention key layers att_key0, att_key1
    std::vector<Affine> L;
    int F = 5;                            // frame-level layers
    int NL = 7;                           // layers in use
    int S0() const { return F; }          // index of the first segment-level layer (tdnn6 in the reference)
    int S1() const { return F + 1; }
    int K0() const { return F + 2; }      // attention key layers
    int K1() const { return F + 3; }
    int amax_a = 1, amax_wt = 0, amax_dz = 0;   // slot ranges inside `amax`, see amax layout below
    bool att = false;
    int v_query = -1;
    float *att_score = nullptr, *att_w = nullptr, *att_dw = nullptr, *att_ds = nullptr;   // [B*T5]
    float* bufA = nullptr;                // d (key input) through the key network, [B*T_pool][width of tdnn(F-1)]
    int min_frames = 15;                  // receptive field of the frame layers
    float* bwd_part = nullptr;            // BN-backward reduction partials written by a data-gradient GEMM epilogue
    int bwd_part_layer = -1, bwd_part_chunks = 0;   // ... for this layer's BN backward (-1: none pending)
    int v_loss_kernel = -1, v_loss_bias = -1, v_ring = -1;
    float* mhe_coef = nullptr;            // [1 + 2*Lout]: g, u, v of the MHE auxiliary loss
    int32_t* mhe_counts = nullptr;        // [N] label histogram
    int c_pad0 = 0;
    int P = 0, Lout = 0, N = 0, ldl = 0;
    // device arena
    char* arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    float *xpad = nullptr, *pool = nullptr, *h7_buf = nullptr, *out_buf = nullptr;
    float *h7 = nullptr, *out = nullptr;   // views of the most recent forward (may alias tdnn7's z / h7)
    float *logits = nullptr, *dlogits = nullptr, *dnorm = nullptr, *row_loss = nullptr;
    float *inv_norm = nullptr, *wn = nullptr, *wnt = nullptr, *dwn = nullptr;
    float *bufD = nullptr, *bufZ[XV_Z_SLOTS] = {}, *d_small0 = nullptr, *d_small1 = nullptr;
    // second stream: weight gradients run beside the data-gradient chain (they only share dz)
    hipStream_t side = nullptr;
    // third stream: the loss head's weight gradient (5 launches, ~0.1 ms alone) starts as soon as dlogits exist and never sits in
    // front of the segment layers' weight gradients on `side` (whose dz slots the main chain is waiting for)
    hipStream_t side2 = nullptr;
    void* ws_side2 = nullptr;
    bool stage_lw = false;        // deferred stage 0: its slice also needs ev_lw
    hipEvent_t ev_dz = nullptr, ev_lw = nullptr, ev_join = nullptr;
    hipEvent_t ev_prep = nullptr, ev_lossprep = nullptr;     // side-stream halves of ensure_weights
    bool prep_pending = false, lossprep_pending = false;
    hipEvent_t ev_comm = nullptr;                 // behind the most recent xv_engine_allreduce on the caller's communication stream
    bool comm_pending = false;
    hipEvent_t ev_stage[XV_BWD_STAGES][2] = {};   // [stage][0 main, 1 side]: that stage's gradients are complete (backward_async)
    bool stage_side[XV_BWD_STAGES] = {};          // the side-stream event of the stage was recorded
    // dz ping-pong state.  ring 0: the frame-level layers' dz (fp16 planes `dzh` in split precision) and, in fp32, every
    // layer's dz (`bufZ`); ring 1 (split precision only): the fp32 dz of the segment-level layers and the attention key
    // gradient in `bufZ` - its own ring, so a frame layer never waits for a segment layer's weight gradient.
    // [measured, same box] giving fp32 mode that second ring as well (it removes a 77 us wait of the last frame layer's BN backward
    // for the slot tdnn7's weight gradient reads) makes the step 0.07 ms SLOWER: the BN backward then runs beside the loss head's
    // side-stream chain and both crawl
    // fp32 mode has a slot per layer when the arena can afford it (`z_private`): no slot is rewritten inside a step, so the data-gradient
    // chain never waits for a weight gradient and the side stream records nothing per layer - every wait / record is a barrier packet
    // that costs the stream it sits on 5-6 us (profiles/r05_event_packets.txt).
    struct ZRing { int cur = 0, n = 2; bool pending[XV_Z_SLOTS] = {}; hipEvent_t ev[XV_Z_SLOTS] = {}; } zr[2];
    int nz = 2;                   // slots of bufZ
    bool z_private = false;       // nz covers every dz of a step
    bool side_dirty = false;      // weight-gradient work is on the side stream since the last join
    int z_taken = 0;              // slots handed to the side stream since the last join
    bool lw_pending = false;      // the loss head's weight gradient (side stream) - it reads no dz buffer, so it has its own event
    bool concurrent = true;
    void* ws_side = nullptr;
    float *scalars = nullptr;   // [0] raw loss, [1] reg loss, [2] grad sumsq
    // segment-level layers in one launch each (xv_skinny.hip) when the batch has <= XV_SEGMENT_MAX_ROWS chunks
    bool sk = true;                 // XV_SEGMENT_FUSED=0 keeps the GEMM / slab-sum / BatchNorm launches apart (A/B, and what B > 128 runs)
    uint32_t* sk_tickets = nullptr; // one per 32 output columns + the loss mean's
    size_t sk_ntickets = 0;
    float* xnorm = nullptr;         // [B] ||out[r]||, written with the loss rows
    float* pool_wpos = nullptr;     // [B][P] share of each chunk's frame weights on ReLU-active frames (pooling forward -> BN backward)
    float* pool_amax = nullptr;     // [B][P] each chunk's largest pooled activation
    bool pool_closed_form = true;   // the last frame layer's BN backward takes its reductions from the pooled statistics (plain ReLU)
    float* lrelu_slope = nullptr;   // network_relu_type lrelu: a constant 0.2 vector as wide as the widest layer
    // split precision state
    bool f16 = false;
    unsigned short* xh = nullptr;             // planes of the (channel-padded) input features
    unsigned short* dzh[2] = {nullptr, nullptr};
    size_t dzh_halfs = 0;                     // halfs per plane of a dz buffer
    uint32_t* amax = nullptr;                 // [AMAX_SLOTS] float bits, see amax_slot()
    bool amax_wt_clean = false, amax_dz_clean = false;   // zeroed by the forward pass's one memset over the whole table
    void* ws = nullptr;
    size_t ws_bytes = 0;
    int32_t* labels_dev = nullptr;   // caller's pointer of the current step
    bool weights_dirty = true;
    const float* pad_src = nullptr;      // engine_forward -> prep_layers: the features whose channel padding rides on the first layer's weight-copy launch
    int pad_rows = 0;
    bool reg_valid = false;
    // state of the most recent forward
    int B = 0, T = 0, training = 0;
    int Tl[XV_MAX_FRAME_LAYERS + 1] = {};   // frames after each frame layer (index 0 = input)
    float lambda = 0.f;
    int with_margin = 1;
    hipStream_t last_stream = nullptr;
    size_t stage_begin[XV_BWD_STAGES], stage_end[XV_BWD_STAGES];
    // scratch of the endpoints that are rebuilt on demand (xv_engine_endpoint: "<layer>_bn", "att_key1_relu"): a buffer of its own, allocated at
    // the first such request - every arena buffer wide enough holds live backward state (a dz slot per layer) between two passes
    float* ep_scratch = nullptr;
    size_t ep_scratch_floats = 0;
};

namespace {

// amax layout (F = frame layers; groups on 16-byte boundaries): 0 input x | amax_a + [0, F): BN+ReLU outputs of tdnn1..F-1 and att_key0 (slot F-1) |
// amax_wt + [0, F+2): weights of tdnn1..F, att_key0/1 (both layouts share a slot) | amax_dz + [0, F+2): dz of the same layers
// (one slot per layer: zeroed once per backward pass, not once per layer)
enum { AMAX_X = 0, AMAX_SLOTS = 64 };
#define AMAX_A (e->amax_a)
#define AMAX_WT (e->amax_wt)
#define AMAX_DZ (e->amax_dz)

// frame-level layers (rows = chunks x frames): tdnn1..F and the attention key layers; the two layers after pooling are segment level
inline bool is_frame(const xv_engine* e, int i) { return i < e->F || i >= e->F + 2; }

float* carve(xv_engine* e, size_t floats) {
    size_t bytes = xv_align(floats * sizeof(float), 256);
    if (e->arena_used + bytes > e->arena_bytes) return nullptr;
    float* p = (float*)(e->arena + e->arena_used);
    e->arena_used += bytes;
    return p;
}

int add_var(xv_engine* e, const std::string& name, std::initializer_list<int> shape, bool trainable) {
    Var v;
    v.name = name;
    v.rank = (int32_t)shape.size();
    v.count = 1;
    int i = 0;
    for (int s : shape) { v.shape[i++] = s; v.count *= (size_t)s; }
    for (; i < 4; ++i) v.shape[i] = 1;
    v.offset = 0;
    v.trainable = trainable;
    e->vars.push_back(v);
    return (int)e->vars.size() - 1;
}

float* vptr(xv_engine* e, int idx) { return e->V + e->vars[idx].offset; }
float* gptr(xv_engine* e, int idx) { return e->G + e->vars[idx].offset; }

// network_relu_type (tdnn.py:24-30): while in scope, the entry points that take a `relu` flag apply y > 0 ? y : slope[c] * y for this
// layer - prelu: slope = the layer's alpha variable (d alpha goes to its gradient slot), lrelu: the constant 0.2 vector (xv_common.h)
struct ActScope {
    ActScope(xv_engine* e, const Affine& a) {
        if (!a.has_relu || e->cfg.relu_type == XV_RELU_RELU) return;
        if (e->cfg.relu_type == XV_RELU_PRELU) xv_set_act_context(vptr(e, a.v_alpha), e->G ? gptr(e, a.v_alpha) : nullptr);
        else xv_set_act_context(e->lrelu_slope, nullptr);
    }
    ~ActScope() { xv_set_act_context(nullptr, nullptr); }
};

void build_variables(xv_engine* e) {
    const xv_config& c = e->cfg;
    const int D = c.feat_dim;
    e->P = c.num_nodes_pooling_layer;
    e->Lout = c.num_nodes_last_layer;
    e->N = c.num_speakers;
    e->att = c.pooling == XV_POOL_SELF_ATTENTION;
    const int F = e->F;
    e->NL = e->att ? F + 4 : F + 2;
    e->L.assign(F + 4, Affine());
    // slot groups start on 16-byte boundaries and are zeroed in multiples of 16 bytes: an unaligned / odd-sized hipMemsetAsync is split
    // into two fill kernels (~5 us each on the stream)
    e->amax_a = 4; e->amax_wt = 4 + (int)xv_align(F, 4); e->amax_dz = e->amax_wt + (int)xv_align(F + 2, 4);
    struct Spec { std::string prefix; const char* kind; const char* scope; int k, cin, cout; bool bn, relu, fused; int in_layer, act; };
    std::vector<Spec> specs;
    {
        int cin = D;
        for (int i = 0; i < F; ++i) {
            const int k = c.num_frame_layers > 0 ? c.frame_context[i] : (i == 0 || i == 1 ? 5 : (i == 2 ? 7 : 1));
            const int w = c.num_frame_layers > 0 ? c.frame_width[i] : (i == F - 1 ? e->P : 512);
            // [TF] rank-4 inputs (the conv layers, tdnn.py:39-93) take the fused BN kernel: SURVEY N4
            specs.push_back({"tdnn" + std::to_string(i + 1), k > 1 ? "conv" : "dense", "", k, cin, w, true, true, k > 1, i - 1, 0});
            cin = w;
        }
    }
    const int Ckey = specs[F - 2].cout;      // the attention key network reads the last-but-one frame layer (tdnn4_relu in the shipped configs)
    specs.push_back({"tdnn" + std::to_string(F + 1), "dense", "", 1, 2 * e->P, 512, true, true, false, -1, 0});
    specs.push_back({"tdnn" + std::to_string(F + 2), "dense", "", 1, 512, e->Lout, !c.last_layer_no_bn, !c.last_layer_linear, false, F, 0});
    // self-attention key network (pooling.py:78-96): dense+bn+relu on the key input, then dense (+tanh)
    specs.push_back({"att_key0", "dense", "attention/att_key0/", 1, Ckey, c.att_key0_nodes, true, true, false, F - 2, 0});
    // last key layer (att_key_network_type): 0 affine, 1 + relu, 3 + tanh as an activation inside the score kernels; 2 = + bn + relu
    specs.push_back({"att_key1", "dense", "attention/att_key1/", 1, c.att_key0_nodes, c.att_key1_nodes, c.att_key_type == 2, c.att_key_type == 2, false,
                     F + 2, c.att_key_type == 2 ? 0 : c.att_key_type});
    // graph-construction order of the reference: the frame layers, the pooling layer's variables, the segment layers
    std::vector<int> order;
    for (int i = 0; i < F; ++i) order.push_back(i);
    order.push_back(F + 2); order.push_back(F + 3); order.push_back(F); order.push_back(F + 1);
    for (int i : order) {
        if (i >= e->NL) continue;
        Affine& a = e->L[i];
        const Spec& s = specs[i];
        a.prefix = s.prefix; a.kind = s.kind; a.scope = s.scope; a.in_layer = s.in_layer; a.act = s.act;
        a.k = s.k; a.c_in = s.cin; a.c_out = s.cout;
        // operand pitch: 16-byte chunks of fp16 planes need multiples of 8 (feature layer 30 -> 32, att_key1 1500 -> 1504)
        a.c_pad = (int)xv_align(s.cin, (i == 0 || (e->f16 && is_frame(e, i))) ? 8 : 4);
        a.o_ld = (int)xv_align(s.cout, 8);
        // The pooled layer's pre-BN tensor and its gradient are the largest tensors of the step and are streamed by HBM-bound kernels (pooling, the
        // pooled BatchNorm backward) and by the GEMMs' LDS-DMA: 1 500 channels = 6 000-byte rows start off the 128-byte grid (nine cache lines
        // per KiB instead of eight).  Plain fp32 path with statistics pooling and a plain ReLU only (the kernels of the other paths take dense rows).
        a.ldz = (i == F - 1 && !e->f16 && !e->att && c.relu_type == XV_RELU_RELU && s.k == 1) ? (int)xv_align(s.cout, 32) : s.cout;
        a.has_bn = s.bn; a.has_relu = s.relu; a.fused_bn = s.fused;
        a.wslot = i < F ? i : i - 2;          // amax slots of the weights: tdnn1..F -> 0..F-1, att_key0/1 -> F, F+1
        a.aslot = i < F ? i : F - 1;          // BN+ReLU output planes: tdnn1..F-1 -> 0..F-2, att_key0 -> F-1
        std::string base = std::string("tdnn/") + s.scope + s.prefix + "_" + s.kind;
        if (s.k > 1) a.v_kernel = add_var(e, base + "/kernel", {1, s.k, s.cin, s.cout}, true);
        else a.v_kernel = add_var(e, base + "/kernel", {s.cin, s.cout}, true);
        a.v_bias = add_var(e, base + "/bias", {s.cout}, true);
        a.v_gamma = a.v_beta = a.v_mmean = a.v_mvar = -1;
        if (s.bn) {
            std::string bn = std::string("tdnn/") + s.scope + s.prefix + "_bn";
            a.v_gamma = add_var(e, bn + "/gamma", {s.cout}, true);
            a.v_beta = add_var(e, bn + "/beta", {s.cout}, true);
            a.v_mmean = add_var(e, bn + "/moving_mean", {s.cout}, false);
            a.v_mvar = add_var(e, bn + "/moving_variance", {s.cout}, false);
        }
        if (c.relu_type == XV_RELU_PRELU && s.relu)      // prelu(x, name): variable_scope("<prefix>_relu") / "alpha" [C], common.py:35-39
            a.v_alpha = add_var(e, std::string("tdnn/") + s.scope + s.prefix + "_relu/alpha", {s.cout}, true);
        if (i == F + 3) e->v_query = add_var(e, "tdnn/attention/query", {1, s.cout}, true);   // [heads, key dim], pooling.py:131
    }
    e->c_pad0 = e->L[0].c_pad;
    if (e->N > 0) {
        e->v_loss_kernel = add_var(e, "softmax/output/kernel", {e->Lout, e->N}, true);
        if (c.loss_kind == XV_LOSS_SOFTMAX) e->v_loss_bias = add_var(e, "softmax/output/bias", {e->N}, true);
        if (c.aux_ring) e->v_ring = add_var(e, "softmax_ringloss/r", {}, true);      // scalar, loss.py:1008-1011
    }
    // offsets: trainable section first (graph order), then non-trainable; 16-byte aligned starts
    size_t off = 0;
    for (auto& v : e->vars) if (v.trainable) { v.offset = off; off += xv_align(v.count, 4); }
    e->n_train = off;
    for (auto& v : e->vars) if (!v.trainable) { v.offset = off; off += xv_align(v.count, 4); }
    e->n_all = off;
    e->n_opt = c.optimizer == 0 ? 0 : (c.optimizer == 1 ? e->n_train : 2 * e->n_train);

    // backward stages -> contiguous gradient ranges (stage 0 finishes the tail of the buffer)
    auto first_off = [&](int layer) { return e->vars[e->L[layer].v_kernel].offset; };
    // stage 0: segment layers + loss | 1: the last two frame layers (+ attention keys, created between them and the segment
    // layers) | 2: the middle frame layers | 3: the first two (reference F = 5: tdnn4-5 | tdnn3 | tdnn1-2)
    const int lo = F >= 4 ? 2 : 1;                                     // first layer of stage 2
    e->stage_begin[0] = first_off(F);     e->stage_end[0] = e->n_train;
    e->stage_begin[1] = first_off(F - 2); e->stage_end[1] = first_off(F);
    e->stage_begin[2] = first_off(lo);    e->stage_end[2] = first_off(F - 2);
    e->stage_begin[3] = 0;                e->stage_end[3] = first_off(lo);
}

int alloc_buffers(xv_engine* e) {
    const xv_config& c = e->cfg;
    const size_t B = c.max_batch, T = c.max_frames;
    const int F = e->F;
    int field = 1;
    for (int i = 0; i < F; ++i) field += e->L[i].k - 1;
    e->min_frames = field;
    XV_REQUIRE(B >= 1 && (int)T >= field, "engine: max_batch >= 1 and max_frames >= %d required (receptive field of the frame layers)", field);
    std::vector<size_t> rows(F + 1);      // rows[i] = chunks x frames entering frame layer i; rows[F] = frames that are pooled
    {
        size_t t_cur = T;
        rows[0] = B * T;
        for (int i = 0; i < F; ++i) { t_cur -= (size_t)(e->L[i].k - 1); rows[i + 1] = B * t_cur; }
        if (c.max_rows > 0) {     // row capacity given: any (chunks, frames) with chunks * frames <= max_rows (one chunk loses the fewest frames)
            XV_REQUIRE(c.max_rows >= field, "engine: max_rows %d is below the receptive field %d", c.max_rows, field);
            for (int i = 0; i <= F; ++i) rows[i] = std::min<size_t>(rows[i], (size_t)c.max_rows);
        }
    }
    const size_t rows_pool = rows[F];
    const int c_key_in = e->L[F - 2].c_out;
    e->ldl = e->N > 0 ? (int)xv_align(e->N, 4) : 0;
    // --- size pass
    size_t need = 0;
    auto want = [&](size_t floats) { need += xv_align(floats * sizeof(float), 256); };
    want(rows[0] * e->c_pad0);
    auto lrows = [&](int i) -> size_t { return i < F ? rows[i + 1] : (i >= F + 2 ? rows_pool : B); };
    for (int i = 0; i < e->NL; ++i) {
        Affine& a = e->L[i];
        size_t r = lrows(i);
        want((size_t)a.c_out * a.k * a.c_pad);                 // wt
        if (a.k > 1) want((size_t)a.c_in * a.k * a.c_out);     // wf
        want(r * a.ldz); want(r * a.c_out);                    // z, a
        want(4 * (size_t)xv_cdiv(r, XV_TILE_M) * a.c_out);     // bn_part
        for (int j = 0; j < 4; ++j) want(a.c_out);
    }
    if (e->f16) {
        want(rows[0] * e->c_pad0);                                              // xh: 2 planes of halfs == 1 float per element
        for (int i = 0; i < e->NL; ++i) {
            if (!is_frame(e, i)) continue;
            Affine& a = e->L[i];
            want((size_t)a.c_out * a.k * a.c_pad);                              // wth
            if (i > 0) want((size_t)a.c_in * a.k * a.o_ld);                     // wfh
            if (i < F - 1 || i == F + 2) want(lrows(i) * a.o_ld);               // ah
            want(a.c_out); want(a.c_out);                                       // zmin, zmax
        }
        want(AMAX_SLOTS);
    }
    if (e->att) { for (int j = 0; j < 4; ++j) want(rows_pool); want(rows_pool * c_key_in); }
    want(B * 2 * e->P); want(B * e->Lout); want(B * e->Lout);
    if (e->N > 0 && c.aux_mhe) { want(1 + 2 * (size_t)e->Lout); want(e->N); }
    if (e->N > 0) {
        want(B * e->ldl); want(B * e->ldl); want(B); want(B);
        want(e->N); want((size_t)e->Lout * e->ldl); want((size_t)e->N * e->Lout); want((size_t)e->Lout * e->ldl);
    }
    // widest frame-level tensor (channels) and the ping-pong buffers of the backward pass: bufD holds d(layer output) /
    // d(layer input) ([rows][c]), bufZ a dz with k-1 zero frames around every chunk
    size_t maxc = 512;
    size_t bufd = 0, bufz = 0, max_pad_rows = 0;
    for (int i = 0; i < e->NL; ++i) {
        if (!is_frame(e, i)) continue;
        const Affine& a = e->L[i];
        maxc = std::max<size_t>(maxc, (size_t)a.c_out);
        maxc = std::max<size_t>(maxc, (size_t)a.c_in);
        const size_t r_out = lrows(i), r_in = i < F ? rows[i] : rows_pool;
        bufd = std::max(bufd, std::max(r_out * a.c_out, r_in * (size_t)a.c_in));
        const size_t padded = r_out + B * 2 * (size_t)(a.k - 1);
        bufz = std::max(bufz, padded * (size_t)a.ldz);
        max_pad_rows = std::max(max_pad_rows, padded);
    }
    // a dz slot per layer (+ the attention key gradient) while that stays below 6 GiB (S5, the extended model at 128 x 400: 4.3 GB); the two-slot
    // ring otherwise - an engine sized for batched extraction (hundreds of thousands of rows, never a backward pass) does not pay for the slots
    e->nz = 2;
    {
        const XvEnv* env = xv_env();
        if (!env) return 2;
        // ... and only an engine with a loss head can run one: a predict-only engine (num_speakers == 0; Trainer.predict_batch builds them with
        // 49 152 rows) keeps the two slots - nine would be 2 GB of arena nothing ever reads
        if (env->dz_slots != 2 && !e->f16 && e->N > 0 && e->NL + 2 <= XV_Z_SLOTS && (size_t)(e->NL + 2) * bufz * sizeof(float) <= ((size_t)6 << 30))
            e->nz = e->NL + 2;
    }
    e->z_private = e->nz > 2;
    e->zr[0].n = e->f16 ? 2 : e->nz;
    e->zr[1].n = 2;
    want(bufd);
    for (int i = 0; i < e->nz; ++i) want(bufz);
    if (e->f16) want((size_t)xv_cdiv(rows[1], XV_TILE_M) * 3 * maxc);
    const size_t dzh_halfs = xv_align(max_pad_rows * (size_t)xv_align(maxc, 8), 8);
    if (e->f16) { want(dzh_halfs); want(dzh_halfs); }
    want(B * (size_t)(2 * e->P > 512 ? 2 * e->P : 512)); want(B * (size_t)(2 * e->P > 512 ? 2 * e->P : 512));
    want(16);
    want(xv_align(maxc, 4) + 4);      // lrelu_slope
    const size_t ntick = xv_skinny_tickets((int)std::max<size_t>(std::max<size_t>(e->N, 2 * (size_t)e->P), std::max<size_t>(maxc, (size_t)e->Lout))) + 8;
    want(ntick); want(B);             // sk_tickets, xnorm
    want(B * (size_t)e->P); want(B * (size_t)e->P);           // pool_wpos, pool_amax
    // GEMM split slabs: weight-gradient partials dominate
    size_t ws = 0;
    for (int i = 0; i < e->NL; ++i) {
        Affine& a = e->L[i];
        size_t r = lrows(i);
        int M = a.k * a.c_pad, Nn = a.c_out;
        size_t s = (size_t)xv_tn_splits(M, Nn, (int)r) * M * Nn * sizeof(float);
        if (s > ws) ws = s;
        if (e->f16 && is_frame(e, i)) {
            s = (size_t)xv_tn16_splits(M, a.o_ld, (int)r) * M * a.o_ld * sizeof(float);
            if (s > ws) ws = s;
        }
    }
    if (e->N > 0) {
        size_t s = (size_t)xv_tn_splits_direct(e->Lout, e->ldl, (int)B) * e->Lout * e->ldl * sizeof(float);
        if (s > ws) ws = s;
        s = (size_t)16 * B * e->ldl * sizeof(float);
        if (s > ws) ws = s;
    }
    size_t opws = xv_op_workspace_bytes((int)rows[1], 2 * e->P, 2 * e->P);
    if (opws > ws) ws = opws;
    if (e->att) {       // xv_att_key_backward: per-chunk partials + the column-sum workspace
        size_t s = 2 * ((size_t)xv_cdiv(rows_pool, 64) * 2 * c.att_key1_nodes * sizeof(float) + xv_op_workspace_bytes((int)rows_pool, c.att_key1_nodes, c.att_key1_nodes)) + 4096;
        if (s > ws) ws = s;
    }
    ws = xv_align(ws, 256);
    need += 3 * ws + 8192;
    XV_CHECK_HIP(hipMalloc((void**)&e->arena, need));
    XV_CHECK_HIP(hipMemset(e->arena, 0, need));
    e->arena_bytes = need;
    e->arena_used = 0;
    // --- carve pass
    e->xpad = carve(e, rows[0] * e->c_pad0);
    for (int i = 0; i < e->NL; ++i) {
        Affine& a = e->L[i];
        size_t r = lrows(i);
        a.wt = carve(e, (size_t)a.c_out * a.k * a.c_pad);
        a.wf = a.k > 1 ? carve(e, (size_t)a.c_in * a.k * a.c_out) : nullptr;
        a.z = carve(e, r * a.ldz);
        a.a = carve(e, r * a.c_out);
        a.bn_part = carve(e, 4 * (size_t)xv_cdiv(r, XV_TILE_M) * a.c_out);
        a.mean = carve(e, a.c_out); a.invstd = carve(e, a.c_out);
        a.scale = carve(e, a.c_out); a.shift = carve(e, a.c_out);
        a.rows = 0;
    }
    if (e->f16) {
        e->xh = (unsigned short*)carve(e, rows[0] * e->c_pad0);
        for (int i = 0; i < e->NL; ++i) {
            if (!is_frame(e, i)) continue;
            Affine& a = e->L[i];
            a.wth_stride = (size_t)a.c_out * a.k * a.c_pad;
            a.wth = (unsigned short*)carve(e, a.wth_stride);
            if (i > 0) {
                a.wfh_stride = (size_t)a.c_in * a.k * a.o_ld;
                a.wfh = (unsigned short*)carve(e, a.wfh_stride);
            }
            if (i < F - 1 || i == F + 2) a.ah = (unsigned short*)carve(e, lrows(i) * a.o_ld);
            a.zmin = carve(e, a.c_out);
            a.zmax = carve(e, a.c_out);
        }
        e->amax = (uint32_t*)carve(e, AMAX_SLOTS);
    }
    if (e->att) {
        e->att_score = carve(e, rows_pool); e->att_w = carve(e, rows_pool); e->att_dw = carve(e, rows_pool); e->att_ds = carve(e, rows_pool);
        e->bufA = carve(e, rows_pool * c_key_in);
    }
    e->pool = carve(e, B * 2 * e->P);
    e->h7_buf = carve(e, B * e->Lout);
    e->out_buf = carve(e, B * e->Lout);
    if (e->N > 0 && c.aux_mhe) {
        e->mhe_coef = carve(e, 1 + 2 * (size_t)e->Lout);
        e->mhe_counts = (int32_t*)carve(e, e->N);
    }
    if (e->N > 0) {
        e->logits = carve(e, B * e->ldl); e->dlogits = carve(e, B * e->ldl);
        e->dnorm = carve(e, B); e->row_loss = carve(e, B);
        e->inv_norm = carve(e, e->N);
        e->wn = carve(e, (size_t)e->Lout * e->ldl);
        e->wnt = carve(e, (size_t)e->N * e->Lout);
        e->dwn = carve(e, (size_t)e->Lout * e->ldl);
    }
    e->bufD = carve(e, bufd);
    for (int i = 0; i < e->nz; ++i) e->bufZ[i] = carve(e, bufz);
    if (e->f16) e->bwd_part = carve(e, (size_t)xv_cdiv(rows[1], XV_TILE_M) * 3 * maxc);
    if (e->f16) {
        e->dzh_halfs = dzh_halfs;
        e->dzh[0] = (unsigned short*)carve(e, dzh_halfs);
        e->dzh[1] = (unsigned short*)carve(e, dzh_halfs);
    }
    size_t small = B * (size_t)(2 * e->P > 512 ? 2 * e->P : 512);
    e->d_small0 = carve(e, small);
    e->d_small1 = carve(e, small);
    e->scalars = carve(e, 16);
    e->lrelu_slope = carve(e, xv_align(maxc, 4) + 4);
    if (c.relu_type == XV_RELU_LRELU) {
        std::vector<float> h(xv_align(maxc, 4) + 4, 0.2f);      // tf.nn.leaky_relu default alpha
        XV_CHECK_HIP(hipMemcpy(e->lrelu_slope, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    e->sk_ntickets = ntick;
    e->sk_tickets = (uint32_t*)carve(e, ntick);       // zero (arena memset); every launch leaves them zero
    e->xnorm = carve(e, B);
    e->pool_wpos = carve(e, B * (size_t)e->P);
    e->pool_amax = carve(e, B * (size_t)e->P);
    {
        const XvEnv* env = xv_env();
        if (!env) return 2;
        e->sk = env->segment_fused != 0;
    }
    e->ws = carve(e, ws / sizeof(float));
    e->ws_side = carve(e, ws / sizeof(float));
    e->ws_side2 = carve(e, ws / sizeof(float));
    e->ws_bytes = ws;
    XV_REQUIRE(e->ws != nullptr && e->ws_side != nullptr && e->scalars != nullptr, "engine: internal arena accounting error");
    {   // lowest priority: the weight-gradient GEMMs are filler work; the small kernels of the critical
        // data-gradient chain must not queue behind their workgroups
        int least = 0, greatest = 0;
        XV_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        XV_CHECK_HIP(hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, least));
        XV_CHECK_HIP(hipStreamCreateWithPriority(&e->side2, hipStreamNonBlocking, least));
    }
    // Events between the engine's own streams order kernels of ONE device: no system-scope fence (cache write-back towards the host and
    // peers) when they are recorded.  Which events carry the system-scope release: ev_join (end of a pass / of a non-deferred stage), ev_stage[][],
    // ev_lw and ev_comm - everything a collective, whose bytes peer GPUs read, may be ordered behind.  Device scope only: ev_dz, the ring
    // events, ev_prep, ev_lossprep (hand-overs between this engine's own kernels).
    const unsigned local = hipEventDisableTiming | hipEventDisableSystemFence;
    XV_CHECK_HIP(hipEventCreateWithFlags(&e->ev_dz, local));
    for (int r = 0; r < 2; ++r)
        for (int i = 0; i < e->zr[r].n; ++i) XV_CHECK_HIP(hipEventCreateWithFlags(&e->zr[r].ev[i], local));
    XV_CHECK_HIP(hipEventCreateWithFlags(&e->ev_lw, hipEventDisableTiming));      // (xv_engine_stage_wait: a collective's stream may wait on it)
    // ev_join is what makes the side streams' weight gradients visible to `s` at the end of a (non-deferred) stage, and a collective enqueued on
    // `s` next hands those bytes to PEER GPUs: it keeps the system-scope release (one packet per pass), like the stage / communication events
    XV_CHECK_HIP(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    XV_CHECK_HIP(hipEventCreateWithFlags(&e->ev_comm, hipEventDisableTiming));
    XV_CHECK_HIP(hipEventCreateWithFlags(&e->ev_prep, local));
    XV_CHECK_HIP(hipEventCreateWithFlags(&e->ev_lossprep, local));
    for (int k = 0; k < XV_BWD_STAGES; ++k)
        for (int j = 0; j < 2; ++j) XV_CHECK_HIP(hipEventCreateWithFlags(&e->ev_stage[k][j], hipEventDisableTiming));
    return 0;
}

// Kernel-layout (and, in split precision, fp16-plane) copies of the weights, rebuilt after every update: one memset + one
// multi-tensor amax + one multi-job layout kernel (+ the loss head's two) instead of ~28 launches.
// With `overlap` (the training forward pass) only the FIRST layer's copies are made on `s`; the other layers' and the loss
// head's go to the side stream behind an event on `s` and are waited for where they are first used (wait_prep before the
// second layer, wait_lossprep before the logits) - they then run under the feature split / first GEMM instead of in front
// of them (58 us of a 5.8 ms fp32 step, 112 us of a 2.7 ms f16x3 step were spent there with the chip otherwise idle).
int wait_prep(xv_engine* e, hipStream_t s) {
    if (e->prep_pending) { XV_CHECK_HIP(hipStreamWaitEvent(s, e->ev_prep, 0)); e->prep_pending = false; }
    return 0;
}
int wait_lossprep(xv_engine* e, hipStream_t s) {
    if (e->lossprep_pending) { XV_CHECK_HIP(hipStreamWaitEvent(s, e->ev_lossprep, 0)); e->lossprep_pending = false; }
    return 0;
}

int prep_layers(xv_engine* e, hipStream_t s, int first, int last) {
    XvPrepJobs J = {};
    XvAmaxJobs A = {};
    for (int i = first; i < last; ++i) {
        Affine& a = e->L[i];
        const float* w = vptr(e, a.v_kernel);
        if (e->f16 && is_frame(e, i)) {
            // fp16 planes scaled by the tensor's own max |w|; the forward and dgrad layouts hold the same values, so one
            // max per layer, taken on the variable itself
            const unsigned* am = e->amax + AMAX_WT + a.wslot;
            XV_REQUIRE(A.n < XV_AMAX_MAX_JOBS, "ensure_weights: too many weight tensors for one amax launch");
            A.x[A.n] = w; A.count[A.n] = (size_t)a.k * a.c_in * a.c_out; A.out[A.n] = e->amax + AMAX_WT + a.wslot; A.n++;
            int rc = xv_prep_add(J, XV_PREP_T16, w, a.k, a.c_in, a.c_out, a.c_pad, a.o_ld, a.wth, (long)a.wth_stride, am);
            if (rc) return rc;
            if (i > 0) {
                rc = xv_prep_add(J, XV_PREP_F16, w, a.k, a.c_in, a.c_out, a.c_pad, a.o_ld, a.wfh, (long)a.wfh_stride, am);
                if (rc) return rc;
            }
        } else {
            int rc = xv_prep_add(J, XV_PREP_T32, w, a.k, a.c_in, a.c_out, a.c_pad, a.c_out, a.wt, 0, nullptr);
            if (rc) return rc;
            if (a.k > 1 && i > 0) {
                rc = xv_prep_add(J, XV_PREP_F32, w, a.k, a.c_in, a.c_out, a.c_pad, a.c_out, a.wf, 0, nullptr);
                if (rc) return rc;
            }
        }
    }
    if (e->pad_src && first == 0 && !e->f16) {      // the step's features ride on the first layer's launch (engine_forward)
        int rc = xv_prep_add(J, XV_PREP_PAD, e->pad_src, 1, e->cfg.feat_dim, e->pad_rows, e->c_pad0, e->c_pad0, e->xpad, 0, nullptr);
        if (rc) return rc;
        e->pad_src = nullptr;
    }
    if (A.n) {
        // one memset over the slot range of these layers (tdnn first..F-1 -> slots first..F-1, key layers -> F, F+1: contiguous)
        unsigned *lo = A.out[0], *hi = A.out[0];
        for (int j = 1; j < A.n; ++j) { lo = std::min(lo, A.out[j]); hi = std::max(hi, A.out[j]); }
        if (e->amax_wt_clean) {
            // the forward pass zeroed the whole table in one memset (ahead of this point on `s`, and of the event the side stream waits for)
        } else if (first == 0 && last == 1) {
            XV_CHECK_HIP(hipMemsetAsync(lo, 0, sizeof(uint32_t), s));                    // layer 0 alone (its neighbours belong to the side-stream half)
        } else {
            XV_CHECK_HIP(hipMemsetAsync(lo, 0, (size_t)(hi - lo + 1) * sizeof(uint32_t), s));
        }
        int rc = xv_launch_amax_multi(s, A);
        if (rc) return rc;
    }
    return xv_launch_weight_prep(s, J);
}

int prep_loss_head(xv_engine* e, hipStream_t s) {
    if (e->N <= 0) return 0;
    return xv_loss_prep_weight(s, vptr(e, e->v_loss_kernel), e->Lout, e->N, e->cfg.loss_kind != XV_LOSS_SOFTMAX, e->inv_norm, e->wn, e->ldl,
                               e->wnt);
}

int ensure_weights(xv_engine* e, hipStream_t s, bool overlap = false) {
    if (!e->weights_dirty) return 0;
    int rc;
    if (overlap && e->concurrent && e->side) {
        rc = prep_layers(e, s, 0, 1);
        if (rc) return rc;
        XV_CHECK_HIP(hipEventRecord(e->ev_dz, s));              // the update that made the copies stale is ahead of this point on `s`
        XV_CHECK_HIP(hipStreamWaitEvent(e->side, e->ev_dz, 0));
        rc = prep_layers(e, e->side, 1, e->NL);
        if (rc) return rc;
        XV_CHECK_HIP(hipEventRecord(e->ev_prep, e->side));
        e->prep_pending = true;
        rc = prep_loss_head(e, e->side);
        if (rc) return rc;
        XV_CHECK_HIP(hipEventRecord(e->ev_lossprep, e->side));
        e->lossprep_pending = true;
    } else {
        rc = prep_layers(e, s, 0, e->NL);
        if (rc) return rc;
        rc = prep_loss_head(e, s);
        if (rc) return rc;
    }
    e->weights_dirty = false;
    return 0;
}

inline int c_relu_type(const xv_engine* e) { return e->cfg.relu_type; }

// BN (+ReLU) forward of one layer given z
int bn_forward(xv_engine* e, hipStream_t s, Affine& a, int rows, bool stats_from_gemm, float* dst_a) {
    const xv_config& c = e->cfg;
    ActScope act(e, a);
    int rc;
    if (e->training && !stats_from_gemm && rows <= XV_BN_SMALL_MAX_ROWS)      // segment-level layers: one launch
        return xv_bn_small_forward(s, a.z, rows, a.c_out, vptr(e, a.v_gamma), vptr(e, a.v_beta), c.bn_epsilon, c.batchnorm_momentum,
                                   a.fused_bn && c.fused_bn_unbiased_moving_var, vptr(e, a.v_mmean), vptr(e, a.v_mvar), a.mean, a.invstd,
                                   a.scale, a.shift, a.has_relu ? 1 : 0, dst_a);
    if (e->training) {
        if (!stats_from_gemm) {
            rc = xv_col_stats(s, a.z, rows, a.c_out, a.ldz, a.bn_part);
            if (rc) return rc;
        }
        rc = xv_bn_finalize(s, a.bn_part, rows, a.c_out, vptr(e, a.v_gamma), vptr(e, a.v_beta), c.bn_epsilon, c.batchnorm_momentum,
                            a.fused_bn && c.fused_bn_unbiased_moving_var, vptr(e, a.v_mmean), vptr(e, a.v_mvar), a.mean, a.invstd,
                            a.scale, a.shift, nullptr, nullptr, nullptr, 1);
    } else {
        rc = xv_bn_inference_scale(s, a.c_out, vptr(e, a.v_gamma), vptr(e, a.v_beta), vptr(e, a.v_mmean), vptr(e, a.v_mvar),
                                   c.bn_epsilon, a.scale, a.shift);
    }
    if (rc) return rc;
    if (!dst_a) return 0;       // the consumer applies scale/shift itself (tdnn5: statistics pooling)
    return xv_bn_apply(s, a.z, rows, a.c_out, a.ldz, a.scale, a.shift, a.has_relu ? 1 : 0, dst_a, a.c_out);
}

}  // namespace

extern "C" int xv_engine_create(const xv_config* cfg, xv_engine** out) {
    XV_REQUIRE(cfg && out, "engine_create: null argument");
    if (!xv_env()) return 2;      // an environment switch this library does not know, or a value it does not understand: refused by name
    XV_REQUIRE(cfg->struct_bytes == (int32_t)sizeof(xv_config),
               "engine_create: xv_config.struct_bytes is %d, this library's xv_config has %d bytes (ABI version %d): rebuild the host against include/xvector_hip.h",
               cfg->struct_bytes, (int)sizeof(xv_config), XV_ABI_VERSION);
    XV_REQUIRE(cfg->feat_dim > 0, "engine_create: feat_dim must be positive");
    XV_REQUIRE(cfg->num_nodes_pooling_layer > 0 && cfg->num_nodes_pooling_layer % 4 == 0,
               "engine_create: num_nodes_pooling_layer must be a positive multiple of 4 (got %d)", cfg->num_nodes_pooling_layer);
    XV_REQUIRE(cfg->num_nodes_last_layer > 0 && cfg->num_nodes_last_layer % 4 == 0,
               "engine_create: num_nodes_last_layer must be a positive multiple of 4 (got %d)", cfg->num_nodes_last_layer);
    XV_REQUIRE(cfg->loss_kind >= XV_LOSS_SOFTMAX && cfg->loss_kind <= XV_LOSS_ARCSOFTMAX, "Not implement loss kind %d", cfg->loss_kind);
    XV_REQUIRE(cfg->optimizer >= 0 && cfg->optimizer <= 2, "Optimizer %d is not supported.", cfg->optimizer);
    if (cfg->loss_kind == XV_LOSS_ASOFTMAX && cfg->num_speakers > 0)
        XV_REQUIRE(cfg->margin_m == 1.f || cfg->margin_m == 2.f || cfg->margin_m == 4.f, "[ERROR] m=%d is not unsupported.", (int)cfg->margin_m);
    XV_REQUIRE(!cfg->feature_norm || cfg->feature_scaling_factor > 0.f, "If feature normalization is applied, scaling factor is necessary.");
    XV_REQUIRE(cfg->precision == XV_PRECISION_F32 || cfg->precision == XV_PRECISION_F16X3, "engine_create: unknown precision %d", cfg->precision);
    XV_REQUIRE(cfg->pooling == XV_POOL_STATISTICS || cfg->pooling == XV_POOL_SELF_ATTENTION, "Not implement pooling kind %d", cfg->pooling);
    if (cfg->pooling == XV_POOL_SELF_ATTENTION) {
        XV_REQUIRE(cfg->att_key0_nodes > 0 && cfg->att_key0_nodes % 4 == 0 && cfg->att_key1_nodes > 0 && cfg->att_key1_nodes % 4 == 0,
                   "engine_create: att_key_num_nodes must be two positive multiples of 4 (got %d, %d)", cfg->att_key0_nodes, cfg->att_key1_nodes);
        XV_REQUIRE(cfg->att_key_type >= 0 && cfg->att_key_type <= 3, "engine_create: att_key_network_type %d is not one of 0..3", cfg->att_key_type);
    }
    XV_REQUIRE(!cfg->aux_mhe || cfg->num_speakers == 0 || cfg->loss_kind != XV_LOSS_SOFTMAX,
               "engine_create: mhe_loss needs a loss with normalised speaker weights (asoftmax / additive margin losses)");
    if (cfg->num_frame_layers != 0) {       // extended frame-layer table (no reference counterpart: tdnn.py hard-codes 5 layers)
        XV_REQUIRE(cfg->num_frame_layers >= 3 && cfg->num_frame_layers <= XV_MAX_FRAME_LAYERS,
                   "engine_create: num_frame_layers must be 0 (the reference's 5) or 3..%d (got %d)", XV_MAX_FRAME_LAYERS, cfg->num_frame_layers);
        for (int i = 0; i < cfg->num_frame_layers; ++i) {
            XV_REQUIRE(cfg->frame_context[i] >= 1 && cfg->frame_context[i] <= 15, "engine_create: frame_context[%d] = %d is outside 1..15", i, cfg->frame_context[i]);
            XV_REQUIRE(cfg->frame_width[i] > 0 && cfg->frame_width[i] % 4 == 0, "engine_create: frame_width[%d] = %d must be a positive multiple of 4", i,
                       cfg->frame_width[i]);
        }
        XV_REQUIRE(cfg->frame_width[cfg->num_frame_layers - 1] == cfg->num_nodes_pooling_layer,
                   "engine_create: the last frame layer is the pooling layer: frame_width[%d] must equal num_nodes_pooling_layer", cfg->num_frame_layers - 1);
    }
    XV_REQUIRE(cfg->relu_type >= XV_RELU_RELU && cfg->relu_type <= XV_RELU_LRELU, "engine_create: unknown network_relu_type code %d", cfg->relu_type);
    XV_REQUIRE(!(cfg->relu_type != XV_RELU_RELU && cfg->pooling == XV_POOL_SELF_ATTENTION && cfg->att_key_type == 1),
               "engine_create: att_key_network_type 1 (affine + relu inside the score kernel) is built for a plain ReLU; use type 0, 2 or 3 with prelu / lrelu");
    xv_engine* e = new xv_engine();
    e->cfg = *cfg;
    e->F = cfg->num_frame_layers > 0 ? cfg->num_frame_layers : 5;
    if (cfg->loss_kind == XV_LOSS_ASOFTMAX && cfg->margin_m == 1.f) {
        // asoftmax with m = 1 returns its plain cross entropy before aux_loss_func is reached (loss.py:110-115): no ring / MHE
        // term, and the variable softmax_ringloss/r is never created
        e->cfg.aux_ring = 0;
        e->cfg.aux_mhe = 0;
    }
    e->f16 = cfg->precision == XV_PRECISION_F16X3;
    build_variables(e);
    int rc = alloc_buffers(e);
    if (rc) { xv_engine_destroy(e); return rc; }
    *out = e;
    return 0;
}

extern "C" void xv_engine_destroy(xv_engine* e) {
    if (!e) return;
    if (e->side) { (void)hipStreamSynchronize(e->side); (void)hipStreamDestroy(e->side); }
    if (e->side2) { (void)hipStreamSynchronize(e->side2); (void)hipStreamDestroy(e->side2); }
    if (e->ev_dz) (void)hipEventDestroy(e->ev_dz);
    for (int r = 0; r < 2; ++r)
        for (int i = 0; i < XV_Z_SLOTS; ++i) if (e->zr[r].ev[i]) (void)hipEventDestroy(e->zr[r].ev[i]);
    if (e->ev_lw) (void)hipEventDestroy(e->ev_lw);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->ev_comm) (void)hipEventDestroy(e->ev_comm);
    if (e->ev_prep) (void)hipEventDestroy(e->ev_prep);
    if (e->ev_lossprep) (void)hipEventDestroy(e->ev_lossprep);
    for (int k = 0; k < XV_BWD_STAGES; ++k)
        for (int j = 0; j < 2; ++j) if (e->ev_stage[k][j]) (void)hipEventDestroy(e->ev_stage[k][j]);

    if (e->arena) (void)hipFree(e->arena);
    if (e->ep_scratch) (void)hipFree(e->ep_scratch);
    delete e;
}

extern "C" int xv_engine_num_variables(const xv_engine* e) { return e ? (int)e->vars.size() : 0; }

extern "C" int xv_engine_variable_info(const xv_engine* e, int index, const char** name, int32_t shape[4], int32_t* rank, size_t* offset,
                                       int32_t* trainable) {
    XV_REQUIRE(e && index >= 0 && index < (int)e->vars.size(), "variable_info: index out of range");
    const Var& v = e->vars[index];
    if (name) *name = v.name.c_str();
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = v.shape[i];
    if (rank) *rank = v.rank;
    if (offset) *offset = v.offset;
    if (trainable) *trainable = v.trainable ? 1 : 0;
    return 0;
}

extern "C" size_t xv_engine_arena_bytes(const xv_engine* e) { return e ? e->arena_bytes : 0; }
extern "C" size_t xv_engine_variables_count(const xv_engine* e) { return e ? e->n_all : 0; }
extern "C" size_t xv_engine_trainable_count(const xv_engine* e) { return e ? e->n_train : 0; }
extern "C" size_t xv_engine_optimizer_state_count(const xv_engine* e) { return e ? e->n_opt : 0; }

extern "C" int xv_engine_bind(xv_engine* e, float* variables, float* grads, float* opt_state) {
    XV_REQUIRE(e && variables, "engine_bind: variables buffer required");
    XV_REQUIRE(((uintptr_t)variables % 16) == 0 && ((uintptr_t)grads % 16) == 0, "engine_bind: buffers must be 16-byte aligned");
    e->V = variables; e->G = grads; e->S = opt_state;
    e->weights_dirty = true;
    return 0;
}

extern "C" int xv_engine_set_concurrency(xv_engine* e, int enabled) {
    XV_REQUIRE(e, "null engine");
    e->concurrent = enabled != 0;
    return 0;
}

extern "C" int xv_engine_invalidate_weights(xv_engine* e) {
    XV_REQUIRE(e, "null engine");
    e->weights_dirty = true;
    return 0;
}

namespace {
int engine_forward(xv_engine* e, void* stream, const float* features, int b, int t, int training, const int32_t* frames);
}
extern "C" int xv_engine_forward(xv_engine* e, void* stream, const float* features, int b, int t, int training) {
    return engine_forward(e, stream, features, b, t, training, nullptr);
}
extern "C" int xv_engine_forward_lengths(xv_engine* e, void* stream, const float* features, int b, int t, const int32_t* frames) {
    XV_REQUIRE(frames, "engine_forward_lengths: the per-chunk frame counts are required");
    return engine_forward(e, stream, features, b, t, 0, frames);
}
namespace {
int engine_forward(xv_engine* e, void* stream, const float* features, int b, int t, int training, const int32_t* frames) {
    XV_REQUIRE(e && e->V, "engine_forward: engine not bound");
    XV_REQUIRE(b >= 1 && b <= e->cfg.max_batch, "engine_forward: batch %d exceeds capacity %d", b, e->cfg.max_batch);
    XV_REQUIRE(t >= e->min_frames && t <= e->cfg.max_frames, "engine_forward: %d frames outside [%d, %d]", t, e->min_frames, e->cfg.max_frames);
    XV_REQUIRE(e->cfg.max_rows <= 0 || (long)b * t <= (long)e->cfg.max_rows, "engine_forward: %d x %d rows exceed the capacity of %d rows", b, t,
               e->cfg.max_rows);
    XV_REQUIRE(!frames || !training, "engine_forward: per-chunk frame counts are an inference-mode input");
    hipStream_t s = (hipStream_t)stream;
    e->last_stream = s;
    e->B = b; e->T = t; e->training = training;
    // training steps only: there xv_engine_loss_forward always follows and picks up the loss head's event
    // split precision: one memset for every max-|x| slot of the step (input, activations, and - when the weight copies are rebuilt, i.e.
    // on every training step - weights and dz) instead of four ~5 us fill launches along the step
    const bool zero_all = e->f16 && e->weights_dirty;
    if (zero_all) {
        XV_CHECK_HIP(hipMemsetAsync(e->amax, 0, AMAX_SLOTS * sizeof(uint32_t), s));
        e->amax_wt_clean = e->amax_dz_clean = true;
    }
    // fp32: the channel padding of the features is one more job of the first layer's weight-copy launch when that launch happens
    // anyway (every training step); otherwise a launch of its own below
    const bool want_pad = !e->f16 && e->weights_dirty;
    e->pad_src = want_pad ? features : nullptr;
    e->pad_rows = b * t;
    int rc = ensure_weights(e, s, training != 0 && e->N > 0);
    e->amax_wt_clean = false;
    const bool padded = want_pad && e->pad_src == nullptr;      // prep_layers took the job
    e->pad_src = nullptr;
    if (rc) return rc;
    int cur_t = t;
    e->Tl[0] = t;
    const int F = e->F;
    if (e->f16) {
        // split precision: every frame-level operand travels as two fp16 planes + a device-side max |x|
        if (!zero_all) XV_CHECK_HIP(hipMemsetAsync(e->amax + AMAX_X, 0, (4 + xv_align(F, 4)) * sizeof(uint32_t), s));      // x and every BN+ReLU output slot
        rc = xv_amax(s, features, (size_t)b * t * e->cfg.feat_dim, e->amax + AMAX_X);
        if (rc) return rc;
        rc = xv_split_planes(s, features, b * t, e->cfg.feat_dim, e->cfg.feat_dim, e->xh, e->c_pad0, (size_t)b * t * e->c_pad0,
                             e->amax + AMAX_X);
        if (rc) return rc;
        const unsigned short* curh = e->xh;
        size_t cur_stride = (size_t)b * t * e->c_pad0;
        const uint32_t* cur_amax = e->amax + AMAX_X;
        for (int i = 0; i < F; ++i) {
            Affine& a = e->L[i];
            int t_out = cur_t - a.k + 1;
            int rows = b * t_out;
            if (i == 1) { rc = wait_prep(e, s); if (rc) return rc; }
            ActScope act(e, a);
            // the epilogue's column min/max are needed in inference too (they fix the next operand's scale)
            rc = xv_affine_forward_f16x3(s, curh, cur_stride, cur_amax, b, cur_t, a.c_pad, a.k, a.wth, a.wth_stride,
                                         e->amax + AMAX_WT + a.wslot, vptr(e, a.v_bias), a.z, a.c_out, a.c_out, a.bn_part);
            if (rc) return rc;
            uint32_t* out_amax = i < F - 1 ? e->amax + AMAX_A + a.aslot : nullptr;
            const xv_config& c = e->cfg;
            if (training) {
                rc = xv_bn_finalize(s, a.bn_part, rows, a.c_out, vptr(e, a.v_gamma), vptr(e, a.v_beta), c.bn_epsilon, c.batchnorm_momentum,
                                    a.fused_bn && c.fused_bn_unbiased_moving_var, vptr(e, a.v_mmean), vptr(e, a.v_mvar), a.mean, a.invstd,
                                    a.scale, a.shift, a.zmin, a.zmax, out_amax, 1);
            } else {
                rc = xv_bn_inference_scale(s, a.c_out, vptr(e, a.v_gamma), vptr(e, a.v_beta), vptr(e, a.v_mmean), vptr(e, a.v_mvar),
                                           c.bn_epsilon, a.scale, a.shift);
                if (rc) return rc;
                if (out_amax) rc = xv_bn_output_range(s, a.bn_part, rows, a.c_out, a.scale, a.shift, 1, a.zmin, a.zmax, out_amax);
            }
            if (rc) return rc;
            if (i < F - 1) {
                rc = xv_bn_apply_split(s, a.z, rows, a.c_out, a.c_out, a.scale, a.shift, 1, out_amax, a.ah, a.o_ld, (size_t)rows * a.o_ld);
                if (rc) return rc;
                curh = a.ah; cur_stride = (size_t)rows * a.o_ld; cur_amax = out_amax;
            }
            a.rows = rows;
            cur_t = t_out;
            e->Tl[i + 1] = t_out;
        }
        if (e->att) {
            // key network on the last-but-one frame layer's output (tdnn4_relu; its planes are still there):
            // att_key0 = dense+bn+relu -> planes, att_key1 = dense
            const xv_config& c = e->cfg;
            Affine &k0 = e->L[e->K0()], &k1 = e->L[e->K1()], &in = e->L[F - 2];
            const int rows = b * cur_t;
            uint32_t* k0_amax = e->amax + AMAX_A + k0.aslot;
            rc = xv_affine_forward_f16x3(s, in.ah, (size_t)rows * in.o_ld, e->amax + AMAX_A + in.aslot, rows, 1, k0.c_pad, 1, k0.wth,
                                         k0.wth_stride, e->amax + AMAX_WT + k0.wslot, vptr(e, k0.v_bias), k0.z, k0.c_out, k0.c_out, k0.bn_part);
            if (rc) return rc;
            ActScope act0(e, k0);
            if (training) {
                rc = xv_bn_finalize(s, k0.bn_part, rows, k0.c_out, vptr(e, k0.v_gamma), vptr(e, k0.v_beta), c.bn_epsilon, c.batchnorm_momentum,
                                    0, vptr(e, k0.v_mmean), vptr(e, k0.v_mvar), k0.mean, k0.invstd, k0.scale, k0.shift, k0.zmin, k0.zmax,
                                    k0_amax, 1);
            } else {
                rc = xv_bn_inference_scale(s, k0.c_out, vptr(e, k0.v_gamma), vptr(e, k0.v_beta), vptr(e, k0.v_mmean), vptr(e, k0.v_mvar),
                                           c.bn_epsilon, k0.scale, k0.shift);
                if (rc) return rc;
                rc = xv_bn_output_range(s, k0.bn_part, rows, k0.c_out, k0.scale, k0.shift, 1, k0.zmin, k0.zmax, k0_amax);
            }
            if (rc) return rc;
            rc = xv_bn_apply_split(s, k0.z, rows, k0.c_out, k0.c_out, k0.scale, k0.shift, 1, k0_amax, k0.ah, k0.o_ld, (size_t)rows * k0.o_ld);
            if (rc) return rc;
            rc = xv_affine_forward_f16x3(s, k0.ah, (size_t)rows * k0.o_ld, k0_amax, rows, 1, k1.c_pad, 1, k1.wth, k1.wth_stride,
                                         e->amax + AMAX_WT + k1.wslot, vptr(e, k1.v_bias), k1.z, k1.c_out, k1.c_out,
                                         k1.has_bn ? k1.bn_part : nullptr);
            if (rc) return rc;
            if (k1.has_bn) {      // att_key_network_type 2: the key is relu(bn(.)), kept in fp32 for the score (no GEMM consumes it)
                ActScope act1(e, k1);
                if (training) {
                    rc = xv_bn_finalize(s, k1.bn_part, rows, k1.c_out, vptr(e, k1.v_gamma), vptr(e, k1.v_beta), c.bn_epsilon,
                                        c.batchnorm_momentum, 0, vptr(e, k1.v_mmean), vptr(e, k1.v_mvar), k1.mean, k1.invstd, k1.scale,
                                        k1.shift, k1.zmin, k1.zmax, nullptr, 1);
                } else {
                    rc = xv_bn_inference_scale(s, k1.c_out, vptr(e, k1.v_gamma), vptr(e, k1.v_beta), vptr(e, k1.v_mmean), vptr(e, k1.v_mvar),
                                               c.bn_epsilon, k1.scale, k1.shift);
                }
                if (rc) return rc;
                rc = xv_bn_apply(s, k1.z, rows, k1.c_out, k1.c_out, k1.scale, k1.shift, 1, k1.a, k1.c_out);
                if (rc) return rc;
            }
            k0.rows = k1.rows = rows;
        }
    } else {
        if (!padded) {
            rc = xv_pad_channels(s, features, b * t, e->cfg.feat_dim, e->xpad, e->c_pad0);
            if (rc) return rc;
        }
        const float* cur = e->xpad;
        for (int i = 0; i < F; ++i) {
            Affine& a = e->L[i];
            int t_out = cur_t - a.k + 1;
            int rows = b * t_out;
            if (i == 1) { rc = wait_prep(e, s); if (rc) return rc; }
            rc = xv_affine_forward(s, cur, b, cur_t, a.c_pad, a.k, a.wt, vptr(e, a.v_bias), a.z, a.c_out, a.ldz,
                                   training ? a.bn_part : nullptr, e->ws, e->ws_bytes);
            if (rc) return rc;
            rc = bn_forward(e, s, a, rows, true, i < F - 1 ? a.a : nullptr);
            if (rc) return rc;
            a.rows = rows;
            cur = a.a; cur_t = t_out;
            e->Tl[i + 1] = t_out;
        }
        if (e->att) {
            Affine &k0 = e->L[e->K0()], &k1 = e->L[e->K1()];
            const int rows = b * cur_t;
            rc = xv_affine_forward(s, e->L[F - 2].a, rows, 1, k0.c_pad, 1, k0.wt, vptr(e, k0.v_bias), k0.z, k0.c_out, k0.c_out,
                                   training ? k0.bn_part : nullptr, e->ws, e->ws_bytes);
            if (rc) return rc;
            rc = bn_forward(e, s, k0, rows, true, k0.a);
            if (rc) return rc;
            rc = xv_affine_forward(s, k0.a, rows, 1, k1.c_pad, 1, k1.wt, vptr(e, k1.v_bias), k1.z, k1.c_out, k1.c_out,
                                   (k1.has_bn && training) ? k1.bn_part : nullptr, e->ws, e->ws_bytes);
            if (rc) return rc;
            if (k1.has_bn) {
                rc = bn_forward(e, s, k1, rows, true, k1.a);
                if (rc) return rc;
            }
            k0.rows = k1.rows = rows;
        }
    }
    const float* frame_w = nullptr;
    if (e->att) {
        // scores = key.query (/ sqrt(dk)), weights = softmax over the frames of each chunk (pooling.py:134-148)
        Affine& k1 = e->L[e->K1()];
        const float scale = e->cfg.att_use_scale ? 1.0f / sqrtf((float)k1.c_out) : 1.0f;
        rc = xv_att_score(s, k1.has_bn ? k1.a : k1.z, b * cur_t, k1.c_out, k1.c_out, k1.act, vptr(e, e->v_query), scale, e->att_score);
        if (rc) return rc;
        rc = xv_softmax_segments_ex(s, e->att_score, b, cur_t, e->att_w, frames, t - cur_t);
        if (rc) return rc;
        frame_w = e->att_w;
    }
    // the last frame layer's BN + ReLU is applied inside the pooling reduction: its [b*t][1500] activation is never written
    {
        ActScope act(e, e->L[F - 1]);
        rc = xv_stat_pool_forward_bn_ex(s, e->L[F - 1].z, b, cur_t, e->P, e->L[F - 1].scale, e->L[F - 1].shift, 1, frame_w, e->pool,
                                        training ? e->pool_wpos : nullptr, e->f16 ? e->pool_amax : nullptr /* bounds |d a| for the dz planes' scale */,
                                        frames, t - cur_t, e->L[F - 1].ldz);
    }
    if (rc) return rc;
    // segment-level layers: dense (+ BatchNorm + activation).  With <= XV_SEGMENT_MAX_ROWS chunks the GEMM, its split-K sum and the
    // training-mode BatchNorm are one launch (xv_skinny.hip); otherwise GEMM + slab sum, then the BatchNorm kernels
    const bool sk = e->sk && b <= XV_SEGMENT_MAX_ROWS;
    auto seg_forward = [&](Affine& a, const float* x, float* dst_a) -> int {
        if (sk) {
            XvSkinny g = {};
            g.A = x; g.lda = a.c_pad; g.Bt = a.wt; g.ldb = a.c_pad; g.M = b; g.N = a.c_out; g.K = a.c_pad;
            g.bias = vptr(e, a.v_bias); g.C = a.z; g.ldc = a.c_out;
            g.ws = e->ws; g.ws_bytes = e->ws_bytes; g.tickets = e->sk_tickets;
            if (a.has_bn && training) {
                const xv_config& c = e->cfg;
                ActScope act(e, a);
                g.epi = XV_SK_BN_FWD;
                g.gamma = vptr(e, a.v_gamma); g.beta = vptr(e, a.v_beta); g.eps = c.bn_epsilon; g.momentum = c.batchnorm_momentum;
                g.unbiased = a.fused_bn && c.fused_bn_unbiased_moving_var; g.mmean = vptr(e, a.v_mmean); g.mvar = vptr(e, a.v_mvar);
                g.mean = a.mean; g.invstd = a.invstd; g.scale = a.scale; g.shift = a.shift;
                g.relu = a.has_relu ? 1 : 0; g.slope = a.has_relu ? xv_act_context().slope : nullptr; g.a_out = dst_a;
                return xv_launch_skinny(s, g);
            }
            g.epi = XV_SK_PLAIN;
            int r = xv_launch_skinny(s, g);
            if (r) return r;
        } else {
            int r = xv_affine_forward(s, x, b, 1, a.c_pad, 1, a.wt, vptr(e, a.v_bias), a.z, a.c_out, a.c_out, nullptr, e->ws, e->ws_bytes);
            if (r) return r;
        }
        return a.has_bn ? bn_forward(e, s, a, b, false, dst_a) : 0;
    };
    Affine& l6 = e->L[e->S0()];
    rc = seg_forward(l6, e->pool, l6.a);
    if (rc) return rc;
    l6.rows = b;
    Affine& l7 = e->L[e->S1()];
    rc = seg_forward(l7, l6.a, e->h7_buf);
    if (rc) return rc;
    l7.rows = b;
    if (l7.has_bn) {
        e->h7 = e->h7_buf;
    } else if (l7.has_relu && c_relu_type(e) != XV_RELU_RELU) {
        ActScope act(e, l7);
        rc = xv_act_small(s, nullptr, l7.z, b, l7.c_out, e->h7_buf);
        if (rc) return rc;
        e->h7 = e->h7_buf;
    } else if (l7.has_relu) {
        rc = xv_relu_backward(s, l7.z, l7.z, (size_t)b * l7.c_out, e->h7_buf);   // z > 0 ? z : 0
        if (rc) return rc;
        e->h7 = e->h7_buf;
    } else {
        e->h7 = l7.z;
    }
    if (e->cfg.feature_norm) {
        rc = xv_l2_scaling_forward(s, e->h7, b, l7.c_out, e->cfg.feature_scaling_factor, e->out_buf);
        if (rc) return rc;
        e->out = e->out_buf;
    } else {
        e->out = e->h7;
    }
    return 0;
}
}  // namespace

extern "C" int xv_engine_loss_forward(xv_engine* e, void* stream, const int32_t* labels, int global_step, int with_margin) {
    XV_REQUIRE(e && e->V && e->N > 0, "engine_loss_forward: engine has no loss head");
    XV_REQUIRE(e->B > 0, "engine_loss_forward: run forward first");
    hipStream_t s = (hipStream_t)stream;
    const xv_config& c = e->cfg;
    const int b = e->B;
    e->labels_dev = (int32_t*)labels;
    e->with_margin = with_margin;
    int rc = ensure_weights(e, s);
    if (rc) return rc;
    rc = wait_prep(e, s);
    if (rc) return rc;
    rc = wait_lossprep(e, s);
    if (rc) return rc;
    if (e->sk && b <= XV_SEGMENT_MAX_ROWS) {
        XvSkinny g = {};
        g.A = e->out; g.lda = e->Lout; g.Bt = e->wnt; g.ldb = e->Lout; g.M = b; g.N = e->N; g.K = e->Lout;
        g.bias = e->v_loss_bias >= 0 ? vptr(e, e->v_loss_bias) : nullptr;
        g.C = e->logits; g.ldc = e->ldl; g.epi = XV_SK_PLAIN;
        g.ws = e->ws; g.ws_bytes = e->ws_bytes; g.tickets = e->sk_tickets;
        rc = xv_launch_skinny(s, g);
    } else {
        XvGemmNT g = {};
        g.A = e->out; g.lda = e->Lout; g.a_rps = 1; g.a_pitch = 1;
        g.Bt = e->wnt; g.ldb = e->Lout;
        g.C = e->logits; g.ldc = e->ldl;
        g.M = b; g.N = e->N; g.K = e->Lout;
        g.bias = e->v_loss_bias >= 0 ? vptr(e, e->v_loss_bias) : nullptr;
        g.ws = e->ws; g.ws_bytes = e->ws_bytes;
        rc = xv_launch_gemm_nt(s, g);
    }
    if (rc) return rc;
    // lambda schedule, loss.py:144-145 (host side: global_step is a fed placeholder, trainer.py:507)
    double lam = (double)c.lambda_base * pow(1.0 + (double)c.lambda_gamma * (double)global_step, -(double)c.lambda_power);
    if (lam < (double)c.lambda_min) lam = (double)c.lambda_min;
    e->lambda = (float)lam;
    int kind = c.loss_kind;
    float m = c.margin_m;
    if (!with_margin && kind != XV_LOSS_SOFTMAX) { kind = XV_LOSS_ASOFTMAX; m = 1.0f; }   // trainer.py:261-271
    // one launch: the rows, ||out[r]|| (divides the ||x|| gradient in backward) and the mean (last ticket of sk_tickets)
    rc = xv_margin_softmax_rows_ex(s, kind, e->logits, b, e->N, e->ldl, e->out, e->Lout, labels, m, e->lambda, e->dlogits, e->dnorm,
                                   e->row_loss, e->scalars + 0, e->xnorm, e->sk_tickets + (e->sk_ntickets - 1));
    if (rc) return rc;
    // auxiliary losses are part of the training loss only (trainer.py:279-289 clears aux_loss_func for validation)
    if (with_margin && c.aux_ring) {
        rc = xv_ring_loss(s, e->out, b, e->Lout, e->Lout, vptr(e, e->v_ring), c.ring_loss_lambda, e->scalars + 0, e->dnorm, e->scalars + 3);
        if (rc) return rc;
    }
    if (with_margin && c.aux_mhe) {
        rc = xv_mhe_loss(s, e->wn, e->Lout, e->N, e->ldl, labels, b, c.mhe_lambda, e->scalars + 0, e->mhe_coef, e->mhe_counts);
        if (rc) return rc;
    }
    e->reg_valid = false;
    return 0;
}

// regularization_loss, trainer.py:357-358.  It does not feed any gradient (the L2 term is added
// analytically in the weight-gradient reduce), so it is only evaluated when the host asks for it
// (the reference fetches it on logging steps only, trainer.py:485-499).
static int compute_reg_loss(xv_engine* e, hipStream_t s) {
    const xv_config& c = e->cfg;
    XV_CHECK_HIP(hipMemsetAsync(e->scalars + 1, 0, sizeof(float), s));
    for (int i = 0; i < e->NL; ++i) {
        int rc = xv_l2_reg_loss(s, vptr(e, e->L[i].v_kernel), e->vars[e->L[i].v_kernel].count, c.weight_l2_regularizer, e->scalars + 1);
        if (rc) return rc;
    }
    if (e->N > 0) {
        float ol2 = c.output_weight_l2_regularizer >= 0.f ? c.output_weight_l2_regularizer : c.weight_l2_regularizer;
        int rc = xv_l2_reg_loss(s, vptr(e, e->v_loss_kernel), e->vars[e->v_loss_kernel].count, ol2, e->scalars + 1);
        if (rc) return rc;
    }
    e->reg_valid = true;
    return 0;
}

namespace {

// Make `waiter` wait for everything enqueued so far on `signaller` (through `ev`).
int chain(hipStream_t signaller, hipStream_t waiter, hipEvent_t ev) {
    XV_CHECK_HIP(hipEventRecord(ev, signaller));
    XV_CHECK_HIP(hipStreamWaitEvent(waiter, ev, 0));
    return 0;
}

// All weight-gradient work enqueued on the side streams so far becomes visible to `s` - through ONE wait on `s`: the side stream is in
// order, so an event recorded on it now covers every dz slot's event, and the loss head's stream is joined into the side stream first.
// (tools/sync_cost_probe.cpp, profiles/r04_sync_cost.txt: in a chain of 10 us kernels a wait for another stream's fresh event costs the
// waiting stream 4 us, a record 3 us, record + the other stream's wait 5.6 us; in the step the difference between three waits and one is
// within the noise of a same-box A/B - as is carrying the hand-over events on the producing kernels' completion signals
// (hipExtLaunchKernel's stopEvent, 1.5 us in the probe), which was built, verified and taken out again.)
int join_side(xv_engine* e, hipStream_t s) {
    bool any = e->side_dirty;
    e->side_dirty = false;
    e->z_taken = 0;
    for (int r = 0; r < 2; ++r)
        for (int i = 0; i < XV_Z_SLOTS; ++i) {
            any = any || e->zr[r].pending[i];
            e->zr[r].pending[i] = false;
        }
    if (e->lw_pending) {
        if (e->side) XV_CHECK_HIP(hipStreamWaitEvent(e->side, e->ev_lw, 0));
        else XV_CHECK_HIP(hipStreamWaitEvent(s, e->ev_lw, 0));
        any = any || e->side;
        e->lw_pending = false;
    }
    if (any && e->side) {
        XV_CHECK_HIP(hipEventRecord(e->ev_join, e->side));
        XV_CHECK_HIP(hipStreamWaitEvent(s, e->ev_join, 0));
    }
    return 0;
}

// backward of one affine(+BN+ReLU) layer.  da: gradient w.r.t. the layer OUTPUT (after BN/ReLU),
// dense [segs*t_out][c_out].  x/t_in: the layer input view.  Writes parameter gradients and, if
// dx != nullptr, the gradient w.r.t. the layer input ([segs*t_in][c_in]).
// The weight/bias gradient only shares dz with the data-gradient chain, so it is enqueued on the
// side stream: its workgroups fill the CUs that the tail of the data-gradient GEMM (and the small
// BN kernels of the next layer) leave idle.  dz ping-pongs between two buffers; a buffer is rewritten
// only after the weight gradient that read it has finished (ZRing).
int layer_backward_f16(xv_engine* e, hipStream_t s, int li, const float* da, int segs, int t_in, float* dx);

// The current slot of the fp32 dz ring, once the weight gradient that last read it (two layers up, side stream) has finished.
float* ring_take(xv_engine* e, hipStream_t s) {
    xv_engine::ZRing& zr = e->zr[e->f16 ? 1 : 0];
    const int zi = zr.cur;
    if (e->z_private && e->z_taken >= zr.n && join_side(e, s)) return nullptr;      // (a caller that never finishes a backward pass)
    if (zr.pending[zi]) {                       // WAR
        if (hipStreamWaitEvent(s, zr.ev[zi], 0) != hipSuccess) return nullptr;
        zr.pending[zi] = false;
    }
    return e->bufZ[zi];
}

// dz of layer `a` (fp32 path) from the gradient w.r.t. its output: BN (+activation) backward, the activation alone, or da itself.
// *ring: dz was written into the ring's current slot (ring_take) - the caller's weight gradient then owns the slot.
int layer_dz(xv_engine* e, hipStream_t s, Affine& a, const float* da, int segs, int t_out, int pad, const float* act_out,
             const float** dz_out, bool* ring) {
    const xv_config& c = e->cfg;
    const int lidx = (int)(&a - &e->L[0]);
    ActScope act(e, a);
    int rc;
    *ring = true;
    float* Z = ring_take(e, s);
    XV_REQUIRE(Z, "engine_backward: waiting for a dz slot failed");
    if (!da) {       // tdnn5: the upstream gradient is the statistics-pooling backward of (pool, d pool)
        XV_REQUIRE(lidx == e->F - 1 && a.has_bn, "engine_backward: only the last frame layer takes its gradient from the pooling layer");
        rc = xv_bn_relu_backward_pooled_ex(s, e->pool, e->d_small0, e->att ? e->att_w : nullptr, e->pool_closed_form ? e->pool_wpos : nullptr, e->B, e->Tl[e->F], a.z, a.c_out,
                                           vptr(e, a.v_gamma), a.mean, a.invstd, a.scale, a.shift, 1, Z, gptr(e, a.v_gamma), gptr(e, a.v_beta),
                                           gptr(e, a.v_bias), e->ws, e->ws_bytes, a.ldz);
    } else if (a.has_bn && pad == 0 && segs * t_out <= XV_BN_SMALL_MAX_ROWS && !is_frame(e, lidx)) {      // segment-level layers: one launch
        rc = xv_bn_small_backward(s, da, a.z, segs * t_out, a.c_out, vptr(e, a.v_gamma), a.mean, a.invstd, a.scale, a.shift,
                                  a.has_relu ? 1 : 0, Z, gptr(e, a.v_gamma), gptr(e, a.v_beta), gptr(e, a.v_bias));
    } else if (a.has_bn) {
        rc = xv_bn_relu_backward(s, da, a.z, segs, t_out, a.c_out, vptr(e, a.v_gamma), a.mean, a.invstd, a.scale, a.shift,
                                 a.has_relu ? 1 : 0, pad, Z, gptr(e, a.v_gamma), gptr(e, a.v_beta), gptr(e, a.v_bias), e->ws, e->ws_bytes);
    } else if (a.has_relu && c.relu_type != XV_RELU_RELU) {       // activation without a BN in front (tdnn7, last_layer_no_bn): needs the pre-activation
        rc = xv_act_small(s, da, a.z, segs * t_out, a.c_out, Z);
    } else if (a.has_relu) {
        rc = xv_relu_backward(s, da, act_out, (size_t)segs * t_out * a.c_out, Z);
    } else {
        *dz_out = da;
        *ring = (da == Z);       // the caller wrote d(output) into the ring's slot itself (attention key gradient)
        return 0;
    }
    *dz_out = Z;
    return rc;
}

// Weight (and, without a BN, bias) gradient of layer `a` from (x, dz).  It only shares dz with the data-gradient chain, so it is
// enqueued on the side stream: its workgroups fill the CUs that the tail of the data-gradient GEMM (and the small BN kernels of
// the next layer) leave idle.  ring: dz is the ring's current slot - it is handed to the side stream and the ring moves on; a slot
// is rewritten only after the weight gradient that read it has finished (ZRing).
// the launches of one layer's weight (and, without a BatchNorm, bias) gradient on `q` with workspace `wws`
int layer_wgrad_on(xv_engine* e, hipStream_t q, void* wws, Affine& a, const float* x, const float* dz, int segs, int t_in, int pad) {
    const xv_config& c = e->cfg;
    const int t_out = t_in - a.k + 1;
    const int seg_pitch = t_out + 2 * pad;
    int rc = xv_affine_wgrad_ld(q, x, segs, t_in, a.c_pad, a.k, a.c_in, dz, a.ldz, seg_pitch, pad, a.c_out, vptr(e, a.v_kernel),
                                c.weight_l2_regularizer, gptr(e, a.v_kernel), wws, e->ws_bytes);
    if (rc) return rc;
    if (!a.has_bn)       // a bias in front of a BN gets its (zero + rounding noise) gradient from the BN backward
        rc = xv_colsum(q, dz, segs * seg_pitch, a.c_out, a.c_out, gptr(e, a.v_bias), wws, e->ws_bytes);
    return rc;
}

int layer_wgrad(xv_engine* e, hipStream_t s, Affine& a, const float* x, const float* dz, int segs, int t_in, int pad, bool ring) {
    xv_engine::ZRing& zr = e->zr[e->f16 ? 1 : 0];
    const int zi = zr.cur;
    const bool concurrent = e->concurrent && ring;   // dz aliasing the caller's buffer: keep everything in order
    hipStream_t ws_stream = concurrent ? e->side : s;
    void* wws = concurrent ? e->ws_side : e->ws;
    int rc;
    if (concurrent) {
        rc = chain(s, e->side, e->ev_dz);
        if (rc) return rc;
    }
    // [measured, round 4, same box, variant builds] the LAST weight-gradient launches of the side stream (tdnn2's; tdnn2-3's; all four) as 768
    // rectangles instead of a full round - so that the BatchNorm backward of tdnn1, which waits 250-300 us for slots beside tdnn2's weight
    // gradient at the very end of the step, finds a free slot per CU: S1 5.22 -> 5.29 / 5.24 / 5.24 ms, 64 x U{200..400} 4.30 -> 4.33 / 4.34 /
    // 4.35 ms.  The full round stays.
    rc = layer_wgrad_on(e, ws_stream, wws, a, x, dz, segs, t_in, pad);
    if (rc) return rc;
    if (concurrent) {
        if (e->z_private) { e->side_dirty = true; ++e->z_taken; }      // the slot is not taken again before the join at the end of the step
        else {
            XV_CHECK_HIP(hipEventRecord(zr.ev[zi], e->side));
            zr.pending[zi] = true;
        }
    }
    if (ring) zr.cur = (zr.cur + 1) % zr.n;
    return 0;
}

int layer_backward(xv_engine* e, hipStream_t s, Affine& a, const float* da, const float* x, int segs, int t_in, float* dx,
                   const float* act_out) {
    const int t_out = t_in - a.k + 1;
    const int pad = (dx && a.k > 1) ? a.k - 1 : 0;
    const int lidx = (int)(&a - &e->L[0]);
    if (e->f16 && is_frame(e, lidx)) return layer_backward_f16(e, s, lidx, da, segs, t_in, dx);
    const float* dz = nullptr;
    bool ring = false;
    int rc = layer_dz(e, s, a, da, segs, t_out, pad, act_out, &dz, &ring);
    if (rc) return rc;
    // the first layer (dx == nullptr) is the end of the chain: nothing is left on `s` to overlap with, and the side stream is still
    // busy with the layer above's weight gradient - its own (small) weight gradient finishes sooner in line on `s`, beside that one
    // (round-2 timeline: 166 us of MFMA-idle tail behind tdnn2's weight gradient: tdnn1's, two slab sums, the update)
    rc = layer_wgrad(e, s, a, x, dz, segs, t_in, pad, ring && dx != nullptr);
    if (rc) return rc;
    if (dx) {
        const float* wf = a.k > 1 ? a.wf : vptr(e, a.v_kernel);
        rc = xv_affine_dgrad_ld(s, dz, a.ldz, segs, t_out, a.c_out, a.k, wf, dx, a.c_in, e->ws, e->ws_bytes);
        if (rc) return rc;
    }
    return 0;
}

// Split-precision backward of a frame layer: dz is written once as fp16 planes (padded layout) and feeds both the
// weight gradient (TN, side stream; A operand = the planes the forward pass already consumed) and the data gradient
// (NT, tap-flipped weight planes).  Same stream / ping-pong protocol as layer_backward.
int layer_backward_f16(xv_engine* e, hipStream_t s, int li, const float* da, int segs, int t_in, float* dx) {
    Affine& a = e->L[li];
    const xv_config& c = e->cfg;
    ActScope act(e, a);
    const int t_out = t_in - a.k + 1;
    const int pad = (dx && a.k > 1) ? a.k - 1 : 0;
    xv_engine::ZRing& zr = e->zr[0];
    const int zi = zr.cur;
    unsigned short* Z = e->dzh[zi];
    uint32_t* zamax = e->amax + AMAX_DZ + a.wslot;
    if (zr.pending[zi]) {
        XV_CHECK_HIP(hipStreamWaitEvent(s, zr.ev[zi], 0));
        zr.pending[zi] = false;
    }
    const int seg_pitch = t_out + 2 * pad;
    const size_t zstride = (size_t)segs * seg_pitch * a.o_ld;
    XV_REQUIRE(zstride <= e->dzh_halfs, "engine_backward: dz plane buffer too small");
    int rc;
    if (da && !a.has_bn) {      // att_key1: `da` already is dz (fp32): planes + the bias gradient straight from it
        XV_REQUIRE(pad == 0 && !a.has_relu, "engine_backward: a frame layer without BN is the attention key layer");
        rc = xv_amax(s, da, (size_t)segs * t_out * a.c_out, zamax);
        if (rc) return rc;
        rc = xv_split_planes(s, da, segs * t_out, a.c_out, a.c_out, Z, a.o_ld, zstride, zamax);
        if (rc) return rc;
        rc = xv_colsum(s, da, segs * t_out, a.c_out, a.c_out, gptr(e, a.v_bias), e->ws, e->ws_bytes);
    } else {
        XvBnBwdSplit x = {};
        x.da = da;
        x.zero_amax = false;           // the dz slots were zeroed at the start of this backward pass
        if (!da) {       // tdnn5: the upstream gradient is the (attention-weighted) pooling backward of (pool, d pool)
            XV_REQUIRE(li == e->F - 1, "engine_backward: only the last frame layer takes its gradient from the pooling layer");
            x.pool_out = e->pool; x.dpool = e->d_small0; x.pool_t = e->Tl[e->F]; x.weights = e->att ? e->att_w : nullptr;
            if (e->pool_closed_form) { x.wpos = e->pool_wpos; x.pamax = e->pool_amax; }
        } else if (e->bwd_part_layer == li && e->bwd_part_chunks == xv_cdiv(segs * t_out, XV_TILE_M)) {
            // the GEMM that produced `da` already reduced it against this layer's z (xv_affine_dgrad_bnstats_f16x3)
            x.ext_part = e->bwd_part; x.ext_chunks = e->bwd_part_chunks;
        }
        rc = xv_bn_relu_backward_split_ex(s, x, a.z, da ? segs : e->B * e->Tl[e->F], da ? t_out : 1, a.c_out, vptr(e, a.v_gamma), a.mean, a.invstd,
                                          a.scale, a.shift, a.zmin, a.zmax, 1, pad, Z, a.o_ld, zstride, zamax, gptr(e, a.v_gamma),
                                          gptr(e, a.v_beta), gptr(e, a.v_bias), e->ws, e->ws_bytes);
    }
    e->bwd_part_layer = -1;
    if (rc) return rc;
    // operand planes of this layer's input: the feature planes for tdnn1, the producing layer's BN+ReLU planes otherwise
    const int in = a.in_layer;
    const unsigned short* xin = in < 0 ? e->xh : e->L[in].ah;
    const int xin_rows = in < 0 ? e->B * e->Tl[0] : e->L[in].rows;
    const uint32_t* xin_amax = in < 0 ? e->amax + AMAX_X : e->amax + AMAX_A + e->L[in].aslot;
    // tdnn1 is the end of the chain: nothing is left on `s` to overlap with, and the side stream is still busy with
    // tdnn2's weight gradient - its own (small) weight gradient finishes sooner in line on `s`
    const bool conc = e->concurrent && li > 0;
    hipStream_t wst = conc ? e->side : s;
    void* wws = conc ? e->ws_side : e->ws;
    if (conc) {
        rc = chain(s, e->side, e->ev_dz);
        if (rc) return rc;
    }
    rc = xv_affine_wgrad_f16x3(wst, xin, (size_t)xin_rows * a.c_pad, xin_amax, segs, t_in, a.c_pad, a.k, a.c_in, Z, zstride, zamax,
                               seg_pitch, pad, a.o_ld, a.c_out, vptr(e, a.v_kernel), c.weight_l2_regularizer, gptr(e, a.v_kernel), wws,
                               e->ws_bytes);
    if (rc) return rc;
    if (conc) {
        XV_CHECK_HIP(hipEventRecord(zr.ev[zi], e->side));
        zr.pending[zi] = true;
    }
    zr.cur ^= 1;
    if (dx) {
        // [measured, round 1] folding the producing layer's BN-backward reductions into this GEMM's epilogue (xv_affine_dgrad_bnstats_f16x3,
        // kept parity-tested at op level) is not used: the epilogue's extra z-tile reads cost each data-gradient GEMM 50-60 us at S1, the
        // reduce kernels they replace 37 us each (3.17 vs 3.03 ms/step)
        rc = xv_affine_dgrad_f16x3(s, Z, zstride, zamax, segs, t_out, a.o_ld, a.k, a.wfh, a.wfh_stride, e->amax + AMAX_WT + a.wslot, dx, a.c_in);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

namespace {
// End of a backward stage.  Joining: `s` waits for the side stream, so the stage's gradients are complete on `s` (a collective
// enqueued on `s` next sees them) - but `s` then also stalls until the weight gradients have drained, which costs the overlap
// of the next stage's data-gradient chain with them (0.2 ms/step at S1).  Deferred: both streams only record an event; whoever
// consumes the slice waits for the pair (xv_engine_stage_wait) on its own stream.
int end_stage(xv_engine* e, hipStream_t s, int stage, bool defer) {
    if (!defer) return join_side(e, s);
    const bool last = stage == XV_BWD_STAGES - 1;
    if (last) {                       // the optimiser step follows on `s`: join here, the event on `s` then covers both streams
        int rc = join_side(e, s);
        if (rc) return rc;
    }
    XV_CHECK_HIP(hipEventRecord(e->ev_stage[stage][0], s));
    if (stage == 0) e->stage_lw = e->lw_pending;      // the loss head's weight gradient (third stream) belongs to this slice
    e->stage_side[stage] = !last && e->concurrent && e->side;
    if (e->stage_side[stage]) XV_CHECK_HIP(hipEventRecord(e->ev_stage[stage][1], e->side));
    return 0;
}
int engine_backward(xv_engine* e, void* stream, int stage, bool defer);
}  // namespace

extern "C" int xv_engine_backward(xv_engine* e, void* stream, int stage) { return engine_backward(e, stream, stage, false); }

extern "C" int xv_engine_backward_async(xv_engine* e, void* stream, int stage) {
    XV_REQUIRE(stage >= 0 && stage < XV_BWD_STAGES, "engine_backward_async: stage %d is not one of 0..%d", stage, XV_BWD_STAGES - 1);
    return engine_backward(e, stream, stage, true);
}

extern "C" int xv_engine_stage_wait(xv_engine* e, void* waiter_stream, int stage) {
    XV_REQUIRE(e && stage >= 0 && stage < XV_BWD_STAGES, "engine_stage_wait: bad arguments");
    hipStream_t w = (hipStream_t)waiter_stream;
    XV_CHECK_HIP(hipStreamWaitEvent(w, e->ev_stage[stage][0], 0));
    if (e->stage_side[stage]) XV_CHECK_HIP(hipStreamWaitEvent(w, e->ev_stage[stage][1], 0));
    if (stage == 0 && e->stage_lw) XV_CHECK_HIP(hipStreamWaitEvent(w, e->ev_lw, 0));
    return 0;
}

namespace {
int engine_backward(xv_engine* e, void* stream, int stage, bool defer) {
    XV_REQUIRE(e && e->V && e->G, "engine_backward: gradient buffer not bound");
    XV_REQUIRE(e->training && e->labels_dev, "engine_backward: needs a training forward + loss_forward first");
    XV_REQUIRE(stage >= -1 && stage < XV_BWD_STAGES, "engine_backward: bad stage %d", stage);
    hipStream_t s = (hipStream_t)stream;
    const xv_config& c = e->cfg;
    const int b = e->B;
    int rc;
    if (stage == -1 || stage == 0) {
        if (e->f16 && !e->amax_dz_clean) XV_CHECK_HIP(hipMemsetAsync(e->amax + AMAX_DZ, 0, xv_align(e->F + 2, 4) * sizeof(uint32_t), s));   // every layer's dz scale slot
        e->amax_dz_clean = false;
        // d wn = out^T . dlogits and the gradient through l2_normalize: on a stream of its own (it only reads dlogits / out / wn, which
        // the main chain never rewrites during backward), started before anything else of the backward pass.  [measured, same box]
        // on the weight-gradient stream, BEHIND the segment layers' weight gradients, it cost fp32 mode 0.2 ms/step (it then ran beside
        // the big data-gradient GEMMs, 10x slower, with the frame layers' weight gradients queued behind it); IN FRONT of them the
        // last frame layer's BN backward waited ~80 us for the dz slot tdnn7's weight gradient still had to read; on its own stream but
        // launched at the END of this stage (it has 3 ms of slack, and d out runs 23 instead of 49 us without it alongside) it again
        // costs fp32 0.24 ms: its many-workgroup TN kernel then competes with the first big data-gradient GEMMs
        // [measured, round 3] started right BEHIND the d-out launch instead of in front of it: no difference (5.36 / 4.42 / 12.70 ms at S1 / 64 x 300 / S5 either way)
        // [measured, round 4, same box, variant builds] started behind the d-POOL launch (the chain d out -> d tdnn6 -> d pool then runs with
        // the chip to itself: d out 23 instead of 49 us, one event instead of three): S1 5.22 -> 5.40 ms, 64 x U{200..400} 4.32 -> 4.41 ms -
        // with the segment layers' weight gradients moved behind it as well 5.41 / 4.44 ms.  Its slab sum and normalisation kernels then run
        // beside the first big GEMMs and crawl (236 / 113 / 247 us for 15 / 33 / 18), and everything queued behind them starts late.
        auto loss_head_wgrad = [&]() -> int {
            int rc = 0;
            hipStream_t ss = e->concurrent ? e->side2 : s;
            void* lws = e->concurrent ? e->ws_side2 : e->ws_side;
            if (e->concurrent) {
                rc = chain(s, ss, e->ev_dz);
                if (rc) return rc;
            }
            XvGemmTN w = {};
            w.A = e->out; w.lda = e->Lout; w.a_rps = b; w.a_pitch = b;
            w.B = e->dlogits; w.ldb = e->ldl; w.b_rps = b; w.b_pitch = b;
            w.M = e->Lout; w.N = e->ldl; w.R = b;
            w.direct = 1;
            w.splits = xv_tn_splits_direct(w.M, w.N, w.R);
            XV_REQUIRE(w.splits == 1 || (size_t)w.splits * w.M * w.N * sizeof(float) <= e->ws_bytes, "engine_backward: workspace too small for the loss weight gradient");
            // unsplit (xv_tn_plan: a short reduction over many tiles): the one "slab" IS d wn [Lout][ldl] - no slab sum
            w.P = w.splits == 1 ? e->dwn : (float*)lws;
            rc = xv_launch_gemm_tn(ss, w);
            if (rc) return rc;
            if (w.splits > 1) {
                rc = xv_launch_wgrad_reduce(ss, w.P, w.splits, 1, e->Lout, e->Lout, e->ldl, e->ldl, nullptr, 0, 0.f, e->dwn, e->ldl);
                if (rc) return rc;
            }
            if (e->with_margin && c.aux_mhe) {
                rc = xv_mhe_add_grad(ss, e->dwn, e->Lout, e->N, e->ldl, e->mhe_coef, e->mhe_counts);
                if (rc) return rc;
            }
            float ol2 = c.output_weight_l2_regularizer >= 0.f ? c.output_weight_l2_regularizer : c.weight_l2_regularizer;
            rc = xv_loss_weight_backward(ss, e->dwn, e->ldl, e->wn, e->ldl, e->inv_norm, vptr(e, e->v_loss_kernel), e->Lout, e->N,
                                         c.loss_kind != XV_LOSS_SOFTMAX, ol2, gptr(e, e->v_loss_kernel), lws, e->ws_bytes);
            if (rc) return rc;
            if (e->v_loss_bias >= 0) {
                rc = xv_colsum(ss, e->dlogits, b, e->N, e->ldl, gptr(e, e->v_loss_bias), lws, e->ws_bytes);
                if (rc) return rc;
            }
            if (e->concurrent) {
                XV_CHECK_HIP(hipEventRecord(e->ev_lw, ss));
                e->lw_pending = true;
            }
            return 0;
        };
        rc = loss_head_wgrad();
        if (rc) return rc;
        // d out = dlogits . wn^T   (pad column of both is zero, so K = ldl is exact), + the gradient through ||out|| (loss.py:122,147).
        // Fused form (xv_skinny.hip): one launch, and with a BatchNorm in tdnn7 and no l2_scaling in between, tdnn7's BN backward too
        Affine &l6 = e->L[e->S0()], &l7 = e->L[e->S1()];
        const bool sk = e->sk && b <= XV_SEGMENT_MAX_ROWS;
        const bool fuse7 = sk && l7.has_bn && !c.feature_norm;
        float* dz7_fused = nullptr;
        if (sk) {
            XvSkinny g = {};
            g.A = e->dlogits; g.lda = e->ldl; g.Bt = e->wn; g.ldb = e->ldl; g.M = b; g.N = e->Lout; g.K = e->ldl;
            g.row_coef = e->dnorm; g.row_norm = e->xnorm; g.X = e->out; g.ldx = e->Lout;
            g.ws = e->ws; g.ws_bytes = e->ws_bytes; g.tickets = e->sk_tickets;
            if (fuse7) {
                dz7_fused = ring_take(e, s);
                XV_REQUIRE(dz7_fused, "engine_backward: waiting for a dz slot failed");
                ActScope act(e, l7);
                const XvActContext ac = xv_act_context();
                g.epi = XV_SK_BN_BWD; g.C = dz7_fused; g.ldc = l7.c_out;
                g.z = l7.z; g.gamma = vptr(e, l7.v_gamma); g.mean = l7.mean; g.invstd = l7.invstd; g.scale = l7.scale; g.shift = l7.shift;
                g.relu = l7.has_relu ? 1 : 0; g.slope = l7.has_relu ? ac.slope : nullptr; g.dalpha = (l7.has_relu && ac.slope) ? ac.dalpha : nullptr;
                g.dgamma = gptr(e, l7.v_gamma); g.dbeta = gptr(e, l7.v_beta); g.dbias = gptr(e, l7.v_bias);
            } else {
                g.epi = XV_SK_PLAIN; g.C = e->d_small0; g.ldc = e->Lout;
            }
            rc = xv_launch_skinny(s, g);
            if (rc) return rc;
        } else {
            XvGemmNT g = {};
            g.A = e->dlogits; g.lda = e->ldl; g.a_rps = 1; g.a_pitch = 1;
            g.Bt = e->wn; g.ldb = e->ldl;
            g.C = e->d_small0; g.ldc = e->Lout;
            g.M = b; g.N = e->Lout; g.K = e->ldl;
            g.ws = e->ws; g.ws_bytes = e->ws_bytes;
            rc = xv_launch_gemm_nt(s, g);
            if (rc) return rc;
            rc = xv_add_norm_grad(s, e->out, e->dnorm, b, e->Lout, e->d_small0);
            if (rc) return rc;
        }
        if (e->v_ring >= 0) {      // d r of the ring loss was evaluated with the loss (0 when the auxiliary loss was off)
            rc = e->with_margin ? xv_copy_2d(s, gptr(e, e->v_ring), 1, e->scalars + 3, 1, 1, 1) : 0;
            if (!e->with_margin) XV_CHECK_HIP(hipMemsetAsync(gptr(e, e->v_ring), 0, sizeof(float), s));
            if (rc) return rc;
        }
        if (!sk) {
            const float* d = e->d_small0;
            if (c.feature_norm) {
                rc = xv_l2_scaling_backward(s, e->h7, d, b, e->Lout, c.feature_scaling_factor, e->d_small1);
                if (rc) return rc;
                d = e->d_small1;
            }
            // tdnn7 -> d a6 (into bufD), tdnn6 -> d pool (into d_small0); the pooling backward itself is evaluated
            // inside tdnn5's BN backward (stage 1) from (pool, d pool): d a5 is never written
            rc = layer_backward(e, s, l7, d, l6.a, b, 1, e->bufD, e->h7);
            if (rc) return rc;
            rc = layer_backward(e, s, l6, e->bufD, e->pool, b, 1, e->d_small0, nullptr);
            if (rc) return rc;
        } else {
            // tdnn7's dz (already there when its BN backward rode on the d-out launch), its weight gradient on the side stream
            const float* dz7 = dz7_fused;
            bool ring7 = true;
            if (!fuse7) {
                const float* d = e->d_small0;
                if (c.feature_norm) {
                    rc = xv_l2_scaling_backward(s, e->h7, d, b, e->Lout, c.feature_scaling_factor, e->d_small1);
                    if (rc) return rc;
                    d = e->d_small1;
                }
                rc = layer_dz(e, s, l7, d, b, 1, 0, e->h7, &dz7, &ring7);
                if (rc) return rc;
            }
            // [measured, round 6, profiles/r06_scheduled_update.txt] both segment layers' weight gradients behind ONE event record (after dz6) on the
            // loss head's stream instead of a record each: S1 +0.3 ... +0.5 %, 64 x U +0.2 %; with no record of their own (launched with the last
            // frame layer's hand-over) +0.8 % / +0.4 % - the packets on the compute stream are not what this chain costs
            rc = layer_wgrad(e, s, l7, l6.a, dz7, b, 1, 0, ring7);
            if (rc) return rc;
            // d a6 = dz7 . W7^T and tdnn6's BatchNorm (+ activation) backward in one launch -> dz6
            float* dz6 = ring_take(e, s);
            XV_REQUIRE(dz6, "engine_backward: waiting for a dz slot failed");
            XV_REQUIRE(l6.has_bn, "engine_backward: the first segment-level layer has a BatchNorm (tdnn.py:147-163)");
            {
                ActScope act(e, l6);
                const XvActContext ac = xv_act_context();
                XvSkinny g = {};
                g.A = dz7; g.lda = l7.c_out; g.Bt = vptr(e, l7.v_kernel); g.ldb = l7.c_out; g.M = b; g.N = l7.c_in; g.K = l7.c_out;
                g.epi = XV_SK_BN_BWD; g.C = dz6; g.ldc = l6.c_out;
                g.z = l6.z; g.gamma = vptr(e, l6.v_gamma); g.mean = l6.mean; g.invstd = l6.invstd; g.scale = l6.scale; g.shift = l6.shift;
                g.relu = l6.has_relu ? 1 : 0; g.slope = l6.has_relu ? ac.slope : nullptr; g.dalpha = (l6.has_relu && ac.slope) ? ac.dalpha : nullptr;
                g.dgamma = gptr(e, l6.v_gamma); g.dbeta = gptr(e, l6.v_beta); g.dbias = gptr(e, l6.v_bias);
                g.ws = e->ws; g.ws_bytes = e->ws_bytes; g.tickets = e->sk_tickets;
                rc = xv_launch_skinny(s, g);
                if (rc) return rc;
            }
            rc = layer_wgrad(e, s, l6, e->pool, dz6, b, 1, 0, true);
            if (rc) return rc;
            // d pool = dz6 . W6^T (into d_small0); the pooling backward itself is evaluated inside the last frame layer's BN backward
            // (stage 1) from (pool, d pool): its d a is never written
            {
                XvSkinny g = {};
                g.A = dz6; g.lda = l6.c_out; g.Bt = vptr(e, l6.v_kernel); g.ldb = l6.c_out; g.M = b; g.N = l6.c_in; g.K = l6.c_out;
                g.epi = XV_SK_PLAIN; g.C = e->d_small0; g.ldc = l6.c_in;
                g.ws = e->ws; g.ws_bytes = e->ws_bytes; g.tickets = e->sk_tickets;
                rc = xv_launch_skinny(s, g);
                if (rc) return rc;
            }
        }
        if (stage == 0) { rc = end_stage(e, s, 0, defer); if (rc) return rc; }
    }
    const int F = e->F;
    const int Tp = e->Tl[F];          // pooled frames
    // backward of frame layer i: a context layer sees chunks of Tl[i] frames, a dense layer one "chunk" per frame
    auto frame_backward = [&](int i, const float* da) -> int {
        Affine& a = e->L[i];
        const float* x = i > 0 ? e->L[i - 1].a : e->xpad;
        float* dx = i > 0 ? e->bufD : nullptr;
        if (a.k > 1) return layer_backward(e, s, a, da, x, b, e->Tl[i], dx, nullptr);
        return layer_backward(e, s, a, da, x, b * e->Tl[i + 1], 1, dx, nullptr);
    };
    const int lo = F >= 4 ? 2 : 1;    // first layer of stage 2 (build_variables: stage ranges)
    if (stage == -1 || stage == 1) {
        if (e->att) {
            // through the attention weights into the key network (pooling.py:134-155): d weights from the pooled statistics,
            // softmax backward, then att_key1 (dense [+ tanh]) and att_key0 (dense + bn + relu) down to the key input (bufA)
            Affine &k0 = e->L[e->K0()], &k1 = e->L[e->K1()];
            const int rows = b * Tp;
            const float scale = c.att_use_scale ? 1.0f / sqrtf((float)k1.c_out) : 1.0f;
            {
                ActScope actv(e, e->L[F - 1]);
                rc = xv_att_pool_backward_weights(s, e->L[F - 1].z, b, Tp, e->P, e->L[F - 1].scale, e->L[F - 1].shift, 1, e->pool, e->d_small0, e->att_dw);
            }
            if (rc) return rc;
            rc = xv_softmax_segments_backward(s, e->att_w, e->att_dw, b, Tp, e->att_ds);
            if (rc) return rc;
            // dzk (fp32) takes the current slot of the fp32 dz ring: a segment-level weight gradient (side stream) may still be
            // reading it.  In split precision its consumer (key1's plane split) runs on `s`, so the slot is not flipped; in
            // fp32 layer_backward() below recognises it as the ring's buffer (dz == Z) and flips the ring itself
            float* dzk = ring_take(e, s);      // (one function hands out every slot: the z_private guard and the ring waits live there)
            XV_REQUIRE(dzk, "engine_backward: waiting for a dz slot failed");
            // with a BN+ReLU key layer (type 2) this is d key (act = 0 on its output) and layer_backward does the BN/ReLU part
            rc = xv_att_key_backward(s, k1.has_bn ? k1.a : k1.z, rows, k1.c_out, k1.act, vptr(e, e->v_query), scale, e->att_ds, dzk,
                                     gptr(e, e->v_query), nullptr, e->ws, e->ws_bytes);
            if (rc) return rc;
            rc = layer_backward(e, s, k1, dzk, k0.a, rows, 1, e->bufD, nullptr);      // -> d att_key0_relu (bufD)
            if (rc) return rc;
            rc = layer_backward(e, s, k0, e->bufD, e->L[F - 2].a, rows, 1, e->bufA, nullptr);  // -> d (key input) through the keys (bufA)
            if (rc) return rc;
        }
        rc = frame_backward(F - 1, nullptr);                                           // last frame layer (da = pooling backward)
        if (rc) return rc;
        if (e->att) rc = xv_add_inplace(s, e->bufD, e->bufA, (size_t)b * Tp * e->L[F - 2].c_out);   // the two paths into the key input
        if (rc) return rc;
        rc = frame_backward(F - 2, e->bufD);
        if (rc) return rc;
        if (stage == 1) { rc = end_stage(e, s, 1, defer); if (rc) return rc; }
    }
    if (stage == -1 || stage == 2) {
        for (int i = F - 3; i >= lo; --i) {
            rc = frame_backward(i, e->bufD);
            if (rc) return rc;
        }
        if (stage == 2) { rc = end_stage(e, s, 2, defer); if (rc) return rc; }
    }
    if (stage == -1 || stage == 3) {
        for (int i = lo - 1; i >= 0; --i) {
            rc = frame_backward(i, e->bufD);
            if (rc) return rc;
        }
        rc = end_stage(e, s, XV_BWD_STAGES - 1, defer);       // end of the backward pass: every gradient is visible to `stream`
        if (rc) return rc;
    }
    return 0;
}
}  // namespace

// ---- gradient exchange for hosts without torch.distributed (the Python host runs the same collective through
// torch.distributed in parallel.py).  RCCL is resolved at first use from the process - the host that created the communicator has it
// loaded, and the communicator must be used with the library that made it - and only then from the default library path: this library
// has no link-time dependency on RCCL and loads on a box without it.
namespace {
typedef int (*rccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*rccl_errstr_fn)(int);
struct Rccl { rccl_allreduce_fn allreduce = nullptr; rccl_errstr_fn errstr = nullptr; bool tried = false; };
Rccl& rccl() {
    static Rccl r;
    if (!r.tried) {
        r.tried = true;
        void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
        void* h = nullptr;
        if (!sym) {
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
            if (h) sym = dlsym(h, "ncclAllReduce");
        }
        r.allreduce = (rccl_allreduce_fn)sym;
        r.errstr = (rccl_errstr_fn)(h ? dlsym(h, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
    }
    return r;
}
}  // namespace

extern "C" int xv_engine_allreduce(xv_engine* e, void* comm_stream, int stage, void* rccl_comm) {
    XV_REQUIRE(e && e->G && rccl_comm && stage >= 0 && stage < XV_BWD_STAGES, "engine_allreduce: bad arguments (stage %d)", stage);
    Rccl& r = rccl();
    XV_REQUIRE(r.allreduce, "engine_allreduce: ncclAllReduce is not available in this process (load RCCL - it made the communicator - first)");
    int rc = xv_engine_stage_wait(e, comm_stream, stage);
    if (rc) return rc;
    const size_t begin = e->stage_begin[stage], count = e->stage_end[stage] - begin;
    if (count > 0) {
        const int nr = r.allreduce(e->G + begin, e->G + begin, count, 7 /* ncclFloat32 */, 0 /* ncclSum */, rccl_comm, (hipStream_t)comm_stream);
        XV_REQUIRE(nr == 0, "engine_allreduce: ncclAllReduce of stage %d (%zu floats) failed: %s", stage, count, r.errstr ? r.errstr(nr) : "?");
    }
    XV_CHECK_HIP(hipEventRecord(e->ev_comm, (hipStream_t)comm_stream));
    e->comm_pending = true;
    return 0;
}

extern "C" int xv_engine_allreduce_wait(xv_engine* e, void* stream) {
    XV_REQUIRE(e, "engine_allreduce_wait: null engine");
    if (e->comm_pending) {
        XV_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, e->ev_comm, 0));
        e->comm_pending = false;
    }
    return 0;
}

extern "C" int xv_engine_stage_grad_range(const xv_engine* e, int stage, size_t* begin, size_t* end) {
    XV_REQUIRE(e && stage >= 0 && stage < XV_BWD_STAGES && begin && end, "stage_grad_range: bad arguments");
    *begin = e->stage_begin[stage];
    *end = e->stage_end[stage];
    return 0;
}

__global__ void clip_scale_kernel(float* __restrict__ g, size_t count, const float* __restrict__ sumsq, float grad_scale, float clip) {
    // tf.clip_by_global_norm: g * clip / max(norm, clip)
    float norm = sqrtf(*sumsq) * grad_scale;
    float k = grad_scale * (clip / fmaxf(norm, clip));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) g[i] *= k;
}

extern "C" int xv_engine_apply(xv_engine* e, void* stream, float lr, float grad_scale, int t) {
    XV_REQUIRE(e && e->V && e->G, "engine_apply: buffers not bound");
    XV_REQUIRE(e->cfg.optimizer == 0 || e->S, "engine_apply: optimiser state buffer not bound");
    hipStream_t s = (hipStream_t)stream;
    const xv_config& c = e->cfg;
    {   // the update rewrites the variables the side-stream halves of ensure_weights read (no-ops after a full step)
        int rcw = wait_prep(e, s);
        if (rcw) return rcw;
        rcw = wait_lossprep(e, s);
        if (rcw) return rcw;
    }
    if (c.clip_gradient_norm > 0.f) {
        XV_CHECK_HIP(hipMemsetAsync(e->scalars + 2, 0, sizeof(float), s));
        int rc = xv_sumsq(s, e->G, e->n_train, e->scalars + 2);
        if (rc) return rc;
        hipLaunchKernelGGL(clip_scale_kernel, dim3(2048), dim3(256), 0, s, e->G, e->n_train, (const float*)(e->scalars + 2), grad_scale,
                           c.clip_gradient_norm);
        XV_LAUNCH_CHECK();
        grad_scale = 1.0f;
    }
    int rc;
    if (c.optimizer == 0) rc = xv_sgd_update(s, e->V, e->G, e->n_train, lr, grad_scale);
    else if (c.optimizer == 1) rc = xv_momentum_update(s, e->V, e->G, e->S, e->n_train, lr, c.momentum, c.use_nesterov, grad_scale);
    else rc = xv_adam_update(s, e->V, e->G, e->S, e->S + e->n_train, e->n_train, lr, 0.9f, 0.999f, 1e-8f, t, grad_scale);
    e->weights_dirty = true;
    e->reg_valid = false;
    return rc;
}

extern "C" int xv_engine_loss_ptrs(xv_engine* e, float** raw_loss, float** reg_loss) {
    XV_REQUIRE(e && e->V, "loss_ptrs: engine not bound");
    if (reg_loss && !e->reg_valid) {
        int rc = compute_reg_loss(e, e->last_stream);
        if (rc) return rc;
    }
    if (raw_loss) *raw_loss = e->scalars + 0;
    if (reg_loss) *reg_loss = e->scalars + 1;
    return 0;
}

// (grows only; the stream is drained before a smaller buffer is freed: a copy of the previous endpoint may still be reading it)
static float* endpoint_scratch(xv_engine* e, size_t floats) {
    if (floats <= e->ep_scratch_floats) return e->ep_scratch;
    if (e->ep_scratch) {
        if (hipStreamSynchronize(e->last_stream) != hipSuccess) { xv_set_error("engine_endpoint: stream synchronisation failed"); return nullptr; }
        (void)hipFree(e->ep_scratch);
        e->ep_scratch = nullptr; e->ep_scratch_floats = 0;
    }
    if (hipMalloc((void**)&e->ep_scratch, floats * sizeof(float)) != hipSuccess) {
        xv_set_error("engine_endpoint: cannot allocate %zu bytes of endpoint scratch", floats * sizeof(float));
        return nullptr;
    }
    e->ep_scratch_floats = floats;
    return e->ep_scratch;
}

extern "C" int xv_engine_endpoint(xv_engine* e, const char* name, float** ptr, int32_t* rows, int32_t* cols, int32_t* ld) {
    XV_REQUIRE(e && name && ptr && rows && cols && ld, "engine_endpoint: null argument");
    XV_REQUIRE(e->B > 0, "engine_endpoint: run forward first");
    std::string n(name);
    auto set = [&](float* p, int r, int c, int l) { *ptr = p; *rows = r; *cols = c; *ld = l; return 0; };
    for (int i = 0; i < e->NL; ++i) {
        Affine& a = e->L[i];
        if (n == a.prefix + "_" + a.kind) return set(a.z, a.rows, a.c_out, a.ldz);
        if (n == a.prefix + "_relu" && a.has_relu) {
            if ((e->f16 && (i < e->F - 1 || i == e->K0())) || i == e->F - 1) {     // not materialised on the hot path (fp16 planes / fused into pooling): rebuild on demand
                ActScope act(e, a);
                int rc = xv_bn_apply(e->last_stream, a.z, a.rows, a.c_out, a.ldz, a.scale, a.shift, 1, a.a, a.c_out);
                if (rc) return rc;
            }
            return set(i == e->S1() ? e->h7 : a.a, a.rows, a.c_out, a.c_out);
        }
        if (n == a.prefix + "_bn" && a.has_bn) {
            if (!a.has_relu) return set(i == e->S1() ? e->h7 : a.a, a.rows, a.c_out, a.c_out);
            // BN output is never materialised on the hot path (fused with ReLU): rebuild on demand
            float* sc = endpoint_scratch(e, (size_t)a.rows * a.c_out);
            if (!sc) return 1;
            int rc = xv_bn_apply(e->last_stream, a.z, a.rows, a.c_out, a.ldz, a.scale, a.shift, 0, sc, a.c_out);
            if (rc) return rc;
            return set(sc, a.rows, a.c_out, a.c_out);
        }
    }
    // debug views of the backward scratch (valid right after backward stage 0)
    if (n == "debug:da5") {     // evaluated on demand with the standalone pooling backward (valid after backward stage 0, before stage 1)
        Affine& a5 = e->L[e->F - 1];
        ActScope act(e, a5);
        int rc = xv_bn_apply(e->last_stream, a5.z, a5.rows, a5.c_out, a5.ldz, a5.scale, a5.shift, 1, a5.a, a5.c_out);
        if (rc) return rc;
        rc = xv_stat_pool_backward(e->last_stream, a5.a, e->pool, e->d_small0, e->B, e->Tl[e->F], e->P, e->bufD);
        if (rc) return rc;
        return set(e->bufD, e->B * e->Tl[e->F], e->P, e->P);
    }
    if (n == "debug:dpool") return set(e->d_small0, e->B, 2 * e->P, 2 * e->P);
    if (n == "attention_weights" && e->att) return set(e->att_w, e->B, e->Tl[e->F], e->Tl[e->F]);     // [b, heads = 1, frames]
    if (n == "att_key1_relu" && e->att && e->L[e->K1()].act == 1) {      // relu key (type 1) lives inside the score kernels: rebuild on demand
        Affine& k1 = e->L[e->K1()];
        float* sc = endpoint_scratch(e, (size_t)k1.rows * k1.c_out);
        if (!sc) return 1;
        int rc = xv_relu_backward(e->last_stream, k1.z, k1.z, (size_t)k1.rows * k1.c_out, sc);      // z > 0 ? z : 0
        if (rc) return rc;
        return set(sc, k1.rows, k1.c_out, k1.c_out);
    }
    if (n == "pooling") return set(e->pool, e->B, 2 * e->P, 2 * e->P);
    if (n == "output") return set(e->out, e->B, e->Lout, e->Lout);
    if (n == "logits" && e->N > 0) return set(e->logits, e->B, e->N, e->ldl);
    xv_set_error("engine_endpoint: unknown endpoint '%s'", name);
    return 3;
}

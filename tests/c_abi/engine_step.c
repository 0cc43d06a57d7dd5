/* Plain C host for the C-ABI of include/xvector_hip.h: no Python, no torch - the FFI boundary as any other language would
 * bind it.  Builds an engine from argv, loads the flat variables / features / labels from a raw little-endian file written
 * by tests/test_gpu_c_abi.py, runs one full optimiser step (forward, loss, backward, apply) on caller-owned hipMalloc
 * buffers and writes raw loss, regularisation loss, the tdnn6_dense embedding and the updated variables back to a file.
 *
 *   engine_step <in.bin> <out.bin> feat_dim num_speakers loss_kind margin_m batch frames precision lr global_step [allreduce]
 *
 * With the optional last argument the backward pass runs in its four stages and every finished gradient slice goes through
 * xv_engine_allreduce on a communication stream of the host's, over an RCCL communicator the host creates (one rank: the one GPU of the
 * test box; the call sequence is the one a multi-GPU host runs, and a sum over one rank must leave every bit where it was).
 *
 * in.bin : float32 variables[variables_count] | float32 features[b*t*d] | int32 labels[b]
 * out.bin: float32 raw_loss | float32 reg_loss | int32 rows | int32 cols | float32 embedding[rows*cols] | float32 variables[...]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "xvector_hip.h"

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)
#define XV_OK(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, xv_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc != 12 && !(argc == 13 && strcmp(argv[12], "allreduce") == 0)) {
        fprintf(stderr, "usage: %s in.bin out.bin feat_dim num_speakers loss_kind margin_m batch frames precision lr global_step [allreduce]\n", argv[0]);
        return 1;
    }
    const int dp = argc == 13;
    xv_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_bytes = (int32_t)sizeof cfg;
    cfg.feat_dim = atoi(argv[3]);
    cfg.num_speakers = atoi(argv[4]);
    cfg.loss_kind = atoi(argv[5]);
    cfg.margin_m = (float)atof(argv[6]);
    const int b = atoi(argv[7]), t = atoi(argv[8]);
    cfg.precision = atoi(argv[9]);
    const float lr = (float)atof(argv[10]);
    const int global_step = atoi(argv[11]);
    cfg.num_nodes_pooling_layer = 1500; cfg.num_nodes_last_layer = 512;
    cfg.last_layer_linear = 1;
    cfg.lambda_min = 0.f; cfg.lambda_base = 1000.f; cfg.lambda_gamma = 1e-4f; cfg.lambda_power = 5.f;
    cfg.weight_l2_regularizer = 1e-2f; cfg.output_weight_l2_regularizer = -1.f;
    cfg.batchnorm_momentum = 0.99f; cfg.bn_epsilon = 1e-3f; cfg.fused_bn_unbiased_moving_var = 1;
    cfg.optimizer = 0; cfg.momentum = 0.9f;
    cfg.max_batch = b; cfg.max_frames = t;
    cfg.pooling = XV_POOL_STATISTICS;

    if (xv_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 4; }
    xv_engine* e = NULL;
    XV_OK(xv_engine_create(&cfg, &e));
    const size_t nv = xv_engine_variables_count(e), nt = xv_engine_trainable_count(e), ns = xv_engine_optimizer_state_count(e);
    const size_t nx = (size_t)b * t * cfg.feat_dim;

    float* h_vars = (float*)malloc(nv * sizeof(float));
    float* h_x = (float*)malloc(nx * sizeof(float));
    int32_t* h_y = (int32_t*)malloc((size_t)b * sizeof(int32_t));
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(h_vars, sizeof(float), nv, f) != nv || fread(h_x, sizeof(float), nx, f) != nx ||
        fread(h_y, sizeof(int32_t), (size_t)b, f) != (size_t)b) {
        fprintf(stderr, "cannot read %s (expects %zu variables)\n", argv[1], nv);
        return 5;
    }
    fclose(f);

    float *d_vars, *d_grads, *d_opt = NULL, *d_x;
    int32_t* d_y;
    HIP_OK(hipMalloc((void**)&d_vars, nv * sizeof(float)));
    HIP_OK(hipMalloc((void**)&d_grads, nt * sizeof(float)));
    if (ns) { HIP_OK(hipMalloc((void**)&d_opt, ns * sizeof(float))); HIP_OK(hipMemset(d_opt, 0, ns * sizeof(float))); }
    HIP_OK(hipMalloc((void**)&d_x, nx * sizeof(float)));
    HIP_OK(hipMalloc((void**)&d_y, (size_t)b * sizeof(int32_t)));
    HIP_OK(hipMemcpy(d_vars, h_vars, nv * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_x, h_x, nx * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_y, h_y, (size_t)b * sizeof(int32_t), hipMemcpyHostToDevice));
    XV_OK(xv_engine_bind(e, d_vars, d_grads, d_opt));

    hipStream_t s;
    HIP_OK(hipStreamCreate(&s));
    XV_OK(xv_engine_forward(e, s, d_x, b, t, 1));
    XV_OK(xv_engine_loss_forward(e, s, d_y, global_step, 1));
    ncclComm_t comm = NULL;
    hipStream_t cs = NULL;
    if (dp) {
        const int dev0 = 0;
        if (ncclCommInitAll(&comm, 1, &dev0) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 7; }
        HIP_OK(hipStreamCreate(&cs));
        for (int stage = 0; stage < XV_BWD_STAGES; ++stage) {
            XV_OK(xv_engine_backward_async(e, s, stage));
            XV_OK(xv_engine_allreduce(e, cs, stage, comm));
        }
        XV_OK(xv_engine_allreduce_wait(e, s));
    } else {
        XV_OK(xv_engine_backward(e, s, -1));
    }
    float *d_raw, *d_reg, *d_emb;
    int32_t rows, cols, ld;
    XV_OK(xv_engine_loss_ptrs(e, &d_raw, &d_reg));
    XV_OK(xv_engine_endpoint(e, "tdnn6_dense", &d_emb, &rows, &cols, &ld));
    float raw, reg;
    float* h_emb = (float*)malloc((size_t)rows * cols * sizeof(float));
    HIP_OK(hipStreamSynchronize(s));
    HIP_OK(hipMemcpy(&raw, d_raw, sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&reg, d_reg, sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy2D(h_emb, (size_t)cols * sizeof(float), d_emb, (size_t)ld * sizeof(float), (size_t)cols * sizeof(float), (size_t)rows,
                       hipMemcpyDeviceToHost));
    XV_OK(xv_engine_apply(e, s, lr, 1.0f, 1));
    HIP_OK(hipStreamSynchronize(s));
    HIP_OK(hipMemcpy(h_vars, d_vars, nv * sizeof(float), hipMemcpyDeviceToHost));

    f = fopen(argv[2], "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", argv[2]); return 6; }
    fwrite(&raw, sizeof(float), 1, f);
    fwrite(&reg, sizeof(float), 1, f);
    fwrite(&rows, sizeof(int32_t), 1, f);
    fwrite(&cols, sizeof(int32_t), 1, f);
    fwrite(h_emb, sizeof(float), (size_t)rows * cols, f);
    fwrite(h_vars, sizeof(float), nv, f);
    fclose(f);
    printf("engine_step: %d variables (%zu floats), loss %.6f reg %.6f, embedding %d x %d\n", xv_engine_num_variables(e), nv, raw, reg, rows, cols);

    xv_engine_destroy(e);
    if (dp) { (void)hipStreamDestroy(cs); (void)ncclCommDestroy(comm); }
    (void)hipStreamDestroy(s);
    (void)hipFree(d_vars); (void)hipFree(d_grads); (void)hipFree(d_opt); (void)hipFree(d_x); (void)hipFree(d_y);
    free(h_vars); free(h_x); free(h_y); free(h_emb);
    return 0;
}

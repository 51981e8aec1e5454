/*
 * dctr.h — C ABI of libdctr_hip.so: the MI355X (gfx950) kernels behind DeepCTR's
 * embedding-lookup + feature-interaction forward path.
 *
 * The reference (shenweichen/DeepCTR, /root/reference) has NO native / FFI layer: every op on this
 * path is a TensorFlow op sequence issued from a Keras `Layer.call`.  Each entry point below therefore
 * replaces one such `call` (or a fused run of them); the reference symbol it stands in for is cited
 * as deepctr/<file>:<lines>.  INTEGRATION.md shows the ctypes stub a DeepCTR maintainer would add.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no C++ types, no exceptions across the boundary.
 *   - every pointer inside an argument struct is a DEVICE pointer on the current HIP device unless
 *     the field comment says "host"; the struct itself is host memory, read during the call only.
 *   - the caller owns every buffer (inputs, outputs, workspace); the library never allocates,
 *     frees or synchronises; work is enqueued on `stream` (a hipStream_t; NULL = default stream).
 *   - return value: 0 = enqueued; <0 = DCTR_E_* argument error (nothing enqueued);
 *     >0 = hipError_t reported by the launch.  dctr_last_error() gives a thread-local message.
 *   - functions are re-entrant and stateless; ordering is by `stream` only.
 *   - fp32 tensors are row-major and contiguous unless a stride field says otherwise.
 */
#ifndef DCTR_H_
#define DCTR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCTR_ABI_VERSION 13

enum {
    DCTR_OK = 0,
    DCTR_E_NULL = -1,        /* required pointer is NULL                    */
    DCTR_E_DIM = -2,         /* size / dimension out of the supported range */
    DCTR_E_ALIGN = -3,       /* pointer or stride not aligned as documented */
    DCTR_E_ENUM = -4,        /* unknown enum value                          */
    DCTR_E_UNSUPPORTED = -5  /* valid request this build does not implement */
};

/* bits OR-ed into the optional device status word by the gather kernels */
enum {
    DCTR_STATUS_INDEX_OOR = 1,   /* an id outside [0, vocabulary): the outputs of that ROW are unspecified (the stand-alone
                                    gather substitutes zeros, the fused dctr_embed_mlp_fwd kernels read table row 0); every
                                    other row is unaffected, no memory outside the tables is touched                  */
    DCTR_STATUS_TIMEOUT = 2      /* a bounded in-kernel wait of the streaming dctr_embed_mlp_fwd kernel expired: the
                                    outputs of that launch are invalid (a defect, reported instead of a hung GPU)     */
};

int dctr_abi_version(void);
const char* dctr_last_error(void);
/* name of the gfx target the kernels were compiled for ("gfx950") */
const char* dctr_target_arch(void);
/* Measurement aid (bench.py): arm a probe for the NEXT kernel launched by this host thread; the launch then
 * carries a start/stop event pair (hipExtLaunchKernelGGL), so dctr_profile_last_ms() returns the duration of
 * that one dispatch as the GPU timestamps it (same quantity as rocprofv3 --kernel-trace), or -1 if none. */
int dctr_profile_next_launch(void);
float dctr_profile_last_ms(void);
/* The same for the next n (<= 256) launches of this host thread — launches that are in flight together on several
 * streams take longer than an isolated one, and that is what rocprofv3 reports for the timed region.
 * dctr_profile_collect waits for them, writes their durations [ms] and returns how many were timed. */
int dctr_wall_clock_khz(void);   /* rate of the device wall clock used by dctr_mlp_args_t.probe */
int dctr_profile_arm(int32_t n);
int dctr_profile_collect(float* ms, int32_t n);

/* ------------------------------------------------------------------------------------------------
 * host input pipeline — what Keras does with the dict of ndarrays handed to predict()/fit()
 * (examples/run_classification_criteo.py:40-50).  HOST pointers, no device work: rows [row_lo, row_lo + n_rows) of every
 * column are converted to dst_kind and written feature-major into dst (column c at dst + c * dst_col_stride elements,
 * normally page-locked memory), by up to n_threads host threads.  Float -> integer conversion truncates (numpy astype).
 * ------------------------------------------------------------------------------------------------ */
enum { DCTR_HOST_I32 = 0, DCTR_HOST_I64 = 1, DCTR_HOST_F32 = 2, DCTR_HOST_F64 = 3 };
typedef struct {
    const void* src;              /* first element of the column (row 0)                                 */
    int64_t stride_bytes;         /* distance between consecutive rows (a column of an [N, k] array: k * itemsize) */
    int32_t kind;                 /* DCTR_HOST_*                                                         */
    int32_t reserved_;
} dctr_host_col_t;
int dctr_host_pack_columns(const dctr_host_col_t* cols, int32_t n_cols, int64_t row_lo, int64_t n_rows, void* dst,
                           int64_t dst_col_stride, int32_t dst_kind, int32_t n_threads);

/* ------------------------------------------------------------------------------------------------
 * a2  Hash.call — deepctr/layers/utils.py:89-112
 *     out = Fingerprint64(decimal_ascii(x)) mod nb  (uint64 modulo, stored as int64),
 *     nb = num_buckets - (mask_zero ? 1 : 0);  mask_zero: out = (out + 1) * (x != 0).
 *     Bit-exact with tf.strings.to_hash_bucket_fast(tf.as_string(x), nb).
 * ------------------------------------------------------------------------------------------------ */
int dctr_hash_bucket_i32(const int32_t* x, int64_t n, int64_t num_buckets, int mask_zero, int64_t* out,
                         void* stream);
int dctr_hash_bucket_i64(const int64_t* x, int64_t n, int64_t num_buckets, int mask_zero, int64_t* out,
                         void* stream);
/* string-dtype features: `bytes` = concatenated UTF-8, offsets[n+1] (int64) delimit element i;
 * mask_zero masks the literal string "0" (utils.py:95,108-110). */
int dctr_hash_bucket_bytes(const uint8_t* bytes, const int64_t* offsets, int64_t n, int64_t num_buckets,
                           int mask_zero, int64_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a2 over the id matrix of a gather (declared after dctr_field_t below): dctr_hash_fields.
 * ------------------------------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------------------------------
 * a3/a4/a6/a7/a8 fused:  embedding_lookup (deepctr/inputs.py:101-117, keras Embedding gather) for
 * every SparseFeat + concat into the DNN-input layout (layers/utils.py:336-346) + Linear.call
 * (layers/utils.py:160-175, the 1-wide `linear0sparse_emb_*` tables of feature_column.py:171-210)
 * + FM.call (layers/interaction.py:588-604), optional in-kernel Hash (a2).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* table;      /* [vocab, dim] fp32; 16-B aligned when dim % 4 == 0                      */
    const float* lin_table;  /* [vocab] fp32 (1-wide linear table) or NULL = no first-order term        */
    int64_t vocab;
    int32_t dim;             /* embedding_dim of this field                                             */
    int32_t out_offset;      /* column of this field inside a dnn_in row; <0: not copied                */
    int32_t in_fm;           /* 1: participates in the FM term                                          */
    int32_t hash_mode;       /* 0: id is the row; 1: Hash(vocab); 2: Hash(vocab, mask_zero=True)        */
    int32_t identity;        /* 1: row = sample index b (field pre-pooled by dctr_embed_pool); id unused */
    int32_t row_pitch;       /* 0: rows are `dim` floats apart and the linear table is a separate [vocab] vector.
                                != 0 (ABI 10, the RECORD form; a multiple of 4, >= dim + 1): row r lies at table + r * row_pitch
                                and its linear entry at lin_table[r * row_pitch] — the caller keeps the first-order weight behind
                                the embedding row (lin_table = table + dim), so that one 128-B line serves both reads of a
                                dim-16 field: 2 x row_pitch = 32 floats.  Honoured by dctr_embed_gather_fm and dctr_embed_mlp_fwd
                                (tile and row-chained kernels; there dim 16 and row_pitch 32 only); every other entry point that
                                takes descriptors rejects args->any_pitch (DCTR_E_UNSUPPORTED)                           */
} dctr_field_t;

enum { DCTR_POOL_SUM = 0, DCTR_POOL_MEAN = 1, DCTR_POOL_MAX = 2 };   /* combiner of a5 (dctr_embed_pool below, dctr_pool_seq_t) */

/* a sequence feature pooled inside dctr_embed_mlp_fwd (dctr_gather_fm_args_t.pools; 32 bytes).  Contract (the descriptors live in
 * device memory, the library cannot check them): int32 ids, idx 8-B aligned, idx_stride even, maxlen even and >= 2, the id matrix and
 * the embedding table below 4 GiB each (32-bit lane offsets from scalar bases). */
typedef struct {
    const void* idx;         /* [rows, idx_stride] int32 ids of the staged data (row 0)                 */
    const int32_t* length;   /* [rows] valid lengths (length_name given) or NULL = mask_zero on id != 0 */
    int64_t idx_stride;      /* elements                                                                */
    int32_t maxlen;          /* T                                                                       */
    int32_t combiner;        /* DCTR_POOL_SUM or DCTR_POOL_MEAN                                         */
} dctr_pool_seq_t;

typedef struct {
    const dctr_field_t* fields;   /* DEVICE array [n_fields]                                            */
    const void* ids;              /* DEVICE id matrix: field j, sample b at ids[j*ids_stride_f + b*ids_stride_b]
                                     (DeepCTR feeds one id column per feature -> natural layout [F, B]);
                                     one row per field, in field order (rows of identity fields unused)   */
    int64_t ids_stride_f;         /* in elements                                                        */
    int64_t ids_stride_b;
    int32_t ids_is_i64;           /* 0: int32 ids, 1: int64 ids                                         */
    int32_t n_fields;
    int32_t max_dim;              /* max over fields of dim (host copy, selects the lane layout)        */
    int32_t all_dim4;             /* 1: every dim % 4 == 0, tables 16-B aligned, out offsets % 4 == 0    */
    int32_t any_hash;             /* 1: some field has hash_mode != 0 (host copy; selects the hashing build)*/
    int32_t n_dense;              /* width of the dense matrix (sum of DenseFeat dimensions), 0 = none   */
    const float* dense;           /* DEVICE [B, dense_stride] dense feature values or NULL              */
    int64_t dense_stride;
    const float* dense_lin_w;     /* DEVICE [n_dense] = Linear.kernel (layers/utils.py:150-158) or NULL */
    int32_t dense_out_offset;     /* first dense column inside a dnn_in row; <0: not copied             */
    int32_t dense_copy_cols;      /* leading dense columns copied into dnn_in (the rest only feed the linear
                                     term: DenseFeat present in linear_feature_columns only); <= n_dense   */
    int64_t batch;
    float* dnn_in;                /* [B, out_stride] or NULL                                            */
    int64_t out_stride;           /* elements; % 4 == 0 when all_dim4                                   */
    float* fm_logit;              /* [B] or NULL:  0.5 * sum_d((sum_f e)^2 - sum_f e^2) over in_fm fields */
    float* lin_logit;             /* [B] or NULL:  sum_f lin_table_f[row] + dense . dense_lin_w         */
    int32_t* status;              /* optional device word, DCTR_STATUS_* bits are OR-ed in              */
    int32_t split_col;            /* dctr_embed_mlp_fwd only; 0 = none.  A dnn_in column (multiple of 64) with:
                                     fields[0 .. split_field) write only columns < split_col, fields[split_field ..)
                                     and the dense passthrough only columns >= split_col.  Lets the fused kernel
                                     build the DNN-input tile in two K-halves (half the LDS, two workgroups per CU). */
    int32_t split_field;
    int32_t uniform_dim;          /* dctr_embed_mlp_fwd only; E > 0 promises: every field has dim == E and out_offset ==
                                     field index * E (the reference's plain DNN input, inputs.py:101-117 + layers/utils.py:
                                     336-346; fields pre-pooled by dctr_embed_pool — identity == 1 — included, see any_identity).
                                     Launches of >= 64 rows per CU then take a persistent kernel: with E in {16, 32} and a DNN
                                     of two or three layers with units[0] in {128, 256}, units[1] in {64, 128}, units[2] in
                                     {64, 128} (ReLU / linear, optional BatchNormalization) the row-chained kernel
                                     (csrc/chain_device.h); else with E in {16, 32, 64} and no identity field the streaming
                                     kernel (LDS-DMA gather ring, 64-row tiles).  0 = unknown / not uniform              */
    int32_t any_identity;         /* 1: some field has identity != 0 (host copy of the descriptors' flags)               */
    int32_t any_pitch;            /* 1: some field has row_pitch != 0 (host copy; ABI 10)                                 */
    int32_t n_pools;              /* dctr_embed_mlp_fwd only (ABI 12); 0 = none.  The LAST n_pools fields are VarLenSparseFeat whose sequences
                                     are pooled INSIDE the launch (inputs.py:120-158, layers/sequence.py:76-106; combiner sum / mean): their
                                     descriptors carry table / lin_table / vocab / dim / out_offset / in_fm as a SparseFeat's would
                                     (identity = 0, hash_mode = 0; their rows of the id matrix are unused), pools[i] the ids of the i-th.
                                     Taken by the row-chained kernel only (launches of >= 64 rows per CU, uniform_dim 16, ReLU / linear DNN
                                     of the 256-128-x family, no identity / hashed / record-form fields, n_pools <= 4 and
                                     pool_pieces + 2 <= n_fields - n_pools: the sequences' rows are requested two positions per
                                     layer-0 step of the SparseFeat k-blocks); everything else answers DCTR_E_UNSUPPORTED — pre-pool with
                                     dctr_embed_pool and hand the fields over as identity fields (ask dctr_mlp_fwd_supported first).
                                     The pooled vectors are bit-identical to dctr_embed_pool's (same order over t, same arithmetic).  */
    const dctr_pool_seq_t* pools; /* DEVICE array [n_pools] (the caller's, like `fields`)                                   */
    int64_t pool_row0;            /* the launch's row 0 is row pool_row0 of the pools' id matrices / length vectors        */
    int32_t pool_pieces;          /* host copy: sum over the pools of maxlen / 2                                          */
    int32_t pool_flags;           /* host copy: bit i = pools[i].combiner == DCTR_POOL_MEAN, bit 4 + i = pools[i].length != NULL */
} dctr_gather_fm_args_t;

int dctr_embed_gather_fm(const dctr_gather_fm_args_t* args, void* stream);

/* a2 Hash.call (deepctr/layers/utils.py:89-112, applied per SparseFeat at inputs.py:108-110) over a whole id matrix in one launch:
 * out[j, b] = Hash(ids[j, b]; vocab_j, mask_zero = (hash_mode_j == 2)) for fields with hash_mode != 0, = ids[j, b] for the others
 * (and identity fields).  `fields` = the gather's DEVICE descriptors; ids / out: [n_fields, batch] with the given strides (elements),
 * int32 or int64 each (a 32-bit `out` needs every hashed vocab < 2^31 and unhashed ids that fit).  The persistent kernels of
 * dctr_embed_mlp_fwd take plain rows: run this first, then the call with descriptors whose hash_mode is 0 and any_hash = 0. */
int dctr_hash_fields(const dctr_field_t* fields, int32_t n_fields, const void* ids, int64_t ids_stride_f, int64_t ids_stride_b,
                     int32_t ids_is_i64, int64_t batch, void* out, int64_t out_stride_f, int32_t out_is_i64, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a5  varlen_embedding_lookup + WeightedSequenceLayer + SequencePoolingLayer
 *     deepctr/inputs.py:120-158, layers/sequence.py:76-106 and :155-183.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const void* idx;         /* [B, T] ids, row stride idx_stride (elements)                            */
    const float* table;      /* [vocab, dim]                                                            */
    const float* lin_table;  /* [vocab] or NULL: also pool the 1-wide linear table into lin_out         */
    const int32_t* length;   /* [B] valid lengths (length_name given) or NULL = mask_zero on idx != 0   */
    const float* weight;     /* [B, T] per-position weights (weight_name) or NULL                       */
    int64_t vocab;
    int64_t idx_stride;
    int64_t batch;
    int32_t idx_is_i64;
    int32_t maxlen;          /* T                                                                       */
    int32_t dim;
    int32_t combiner;        /* DCTR_POOL_*                                                             */
    int32_t weight_norm;     /* softmax-normalise the weights over valid positions (sequence.py:170-177)*/
    int32_t hash_mode;       /* 0 none, 2 = Hash(vocab, mask_zero=True) (inputs.py:125-126)             */
    float* out;              /* [B, out_stride] pooled vectors                                          */
    int64_t out_stride;
    float* lin_out;          /* [B] pooled 1-wide term or NULL                                          */
    int32_t* status;
} dctr_pool_args_t;

int dctr_embed_pool(const dctr_pool_args_t* args, void* stream);

/* stand-alone WeightedSequenceLayer.call (layers/sequence.py:155-183) on a materialised seq [B,T,dim]:
 * out = seq * w, w = weight masked by `mask` [B,T] bytes or `length` [B] (0 outside; softmax over T of the
 * -2^32+1 padded weights when weight_norm). */
int dctr_seq_weight_fwd(const float* seq, const float* weight, const uint8_t* mask, const int32_t* length,
                        int64_t batch, int32_t maxlen, int32_t dim, int32_t weight_norm, float* out, void* stream);

/* plain per-position lookup [B,T] -> [B,T,dim] (+ mask byte per position), used for DIN keys
 * (deepctr/models/sequence/din.py:68-69) and by the eager `embedding_lookup` API. */
typedef struct {
    const void* idx;
    const float* table;
    int64_t vocab;
    int64_t n;               /* number of ids (B*T)                                                     */
    int32_t idx_is_i64;
    int32_t dim;
    int32_t hash_mode;
    int32_t pad_;
    float* out;              /* [n, out_stride] (out_stride >= dim): lets callers write into a concat   */
    int64_t out_stride;
    uint8_t* mask;           /* [n] (post-hash idx != 0) or NULL                                        */
    int32_t* status;
} dctr_lookup_args_t;

int dctr_embed_lookup(const dctr_lookup_args_t* args, void* stream);
/* up to 8 lookups in ONE launch (DIN's query features + behaviour sequences, din.py:62-76).  A lookup with mask != NULL
 * additionally ANDs (id != 0) of n_extra (<= 4) further id arrays of the same length into its mask (HOST arrays of
 * DEVICE pointers / int64 flags): the conjunction of the mask_zero sequences' masks. */
int dctr_embed_lookup_multi(const dctr_lookup_args_t* args, int32_t n_lookups, const void* const* extra_mask_idx,
                            const int32_t* extra_is_i64, int32_t n_extra, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a8  FM.call stand-alone — deepctr/layers/interaction.py:588-604.   x [B,F,E] (sample stride x_stride) -> y [B]
 * ------------------------------------------------------------------------------------------------ */
int dctr_fm_fwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim, float* y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a9  CrossNet.call — deepctr/layers/interaction.py:405-424
 *     vector: x_{l+1} = x0 * (x_l . w_l) + b_l + x_l ;  matrix: x_{l+1} = x0 .* (W_l x_l + b_l) + x_l
 *     kernels: [layers, d] (vector) or [layers, d, d] (matrix, W[i][j] row-major); bias [layers, d].
 * ------------------------------------------------------------------------------------------------ */
enum { DCTR_CROSS_VECTOR = 0, DCTR_CROSS_MATRIX = 1 };
/* Device scratch the matrix form needs when dim % 4 != 0 or `kernels` is not 16-B aligned (the W rows are re-packed to
 * a 16-B aligned stride for the dwordx4 weight loads); 0 otherwise. */
size_t dctr_crossnet_workspace_bytes(int32_t dim, int32_t layers, int32_t mode, const float* kernels);
int dctr_crossnet_fwd(const float* x, int64_t batch, int32_t dim, int64_t x_stride, const float* kernels,
                      const float* bias, int32_t layers, int32_t mode, float* y, int64_t y_stride, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The same with the branch's share of the model's head fused in (ABI 6; models/dcn.py:61-64: Dense(1, use_bias=False) over
 * Concatenate([cross_out, deep_out]) = cross_out . kernel[:d] + deep_out . kernel[d:]):  logit[b] = x_L[b, :] . head_w  is written
 * beside the layer output (y != NULL) or instead of it (y == NULL: the [B, d] output never goes to HBM); the caller hands `logit` to
 * the DNN's fused head as one of dctr_mlp_args_t.add.  workspace_ready: the workspace already holds the re-packed rows of THESE
 * kernels (the re-pack launch is skipped; inference with fixed weights). */
typedef struct {
    const float* x;               /* [B, x_stride]                                                        */
    int64_t batch;
    int64_t x_stride;
    int32_t dim;
    int32_t layers;
    int32_t mode;                 /* DCTR_CROSS_VECTOR | DCTR_CROSS_MATRIX                                */
    int32_t workspace_ready;
    const float* kernels;
    const float* bias;
    float* y;                     /* [B, y_stride] or NULL (then head_w / logit are required)             */
    int64_t y_stride;
    void* workspace;              /* dctr_crossnet_workspace_bytes() bytes, 16-B aligned (or NULL when 0) */
    size_t workspace_bytes;
    const float* head_w;          /* [dim] or NULL                                                        */
    float* logit;                 /* [B] written, or NULL                                                 */
    float* save_u;                /* training, matrix form: NULL, or [layers, B, dim] — u_l = W_l x_l (bias excluded) written for
                                     dctr_crossnet_bwd_args_t.saved_u                                                       */
    float* save_x;                /* ... and [layers - 1, B, dim] — x_1 .. x_{L-1} for saved_x (NULL with one layer)        */
} dctr_crossnet_args_t;
int dctr_crossnet_head_fwd(const dctr_crossnet_args_t* args, void* stream);
/* ABI 8 — the same over the embeddings of a gather (models/dcn.py:48-66: dnn_input = combined_dnn_input(sparse embeddings, dense values)
 * -> CrossNet -> its share of Dense(1)): matrix parameterization; the workgroup's [64, dim] tile of the DNN input is read from the
 * embedding tables and the dense matrix inside the kernel (gather: the dctr_embed_gather_fm arguments of the same batch — fields / ids /
 * strides / dense / status are used; every field a plain lookup of width uniform_dim % 4 == 0 at column field * uniform_dim, the dense
 * columns behind them; args->x / x_stride are ignored), so the DNN input never has to exist in HBM.  Launches of >= 64 rows per CU with
 * dim <= 512 (the 64-row kernel whose layer outputs wait in registers); anything else returns DCTR_E_UNSUPPORTED — use
 * dctr_embed_gather_fm + dctr_crossnet_head_fwd.  Inference only (save_u / save_x must be NULL). */
int dctr_crossnet_gather_head_fwd(const dctr_crossnet_args_t* args, const dctr_gather_fm_args_t* gather, void* stream);
/* ABI 13 — would dctr_crossnet_head_fwd / dctr_crossnet_fwd (gather == NULL) or dctr_crossnet_gather_head_fwd (gather != NULL) take these
 * arguments?  Runs every shape check and kernel-shape decision of the call (rows per workgroup by batch and dim, LDS) without launching;
 * device pointers are not looked at (save_u non-NULL = "the training forward"; of `gather` the summary fields n_fields / batch /
 * uniform_dim / all_dim4 / any_hash / any_identity / any_pitch / ids_stride_b / dense_copy_cols / dense_out_offset).  1 = yes, 0 = no
 * (dctr_last_error() carries the reason).  A matrix form wider than the kernels hold runs as dctr_sgemm + dctr_crossnet_matrix_step. */
int dctr_crossnet_fwd_supported(const dctr_crossnet_args_t* args, const dctr_gather_fm_args_t* gather);
/* ABI 11 — the elementwise half of ONE matrix-form layer (interaction.py:416-420) for inputs wider than the kernels above hold on chip
 * (their [16, dim] tiles of x_0 / x_l / x_{l+1} live in LDS: dim <= ~800; wider calls return DCTR_E_UNSUPPORTED):
 *     x_next[b, c] = x0[b, c] * (u[b, c] + bias[c]) + xl[b, c],     u = x_l W_l^T from dctr_sgemm ([B, dim], contiguous).
 * x_next may alias xl (not x0 unless xl == x0 is no longer needed). */
int dctr_crossnet_matrix_step(const float* x0, int64_t x0_stride, const float* xl, int64_t xl_stride, const float* u, const float* bias,
                              int64_t batch, int32_t dim, float* x_next, int64_t x_next_stride, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a10 CIN.call — deepctr/layers/interaction.py:277-325   (outer product + 1x1 conv on f32 MFMA)
 *     x [B,F0,D];  filters[k]: [F0*F_k, H_k] row-major (the reference's [1,F0*F_k,H_k] squeezed,
 *     row index i*F_k + j);  bias[k]: [H_k];  out [B, featuremap_num] (already summed over D).
 * ------------------------------------------------------------------------------------------------ */
enum { DCTR_ACT_LINEAR = 0, DCTR_ACT_RELU = 1, DCTR_ACT_SIGMOID = 2, DCTR_ACT_TANH = 3, DCTR_ACT_DICE = 4 };
typedef struct {
    const float* x;               /* sample b at x + b * x_stride, then [F0, D] row-major */
    int64_t batch;
    int64_t x_stride;             /* elements between samples (>= F0*D): lets CIN read the dnn_in concat in place */
    int32_t fields;               /* F0 */
    int32_t dim;                  /* D  */
    int32_t n_layers;
    int32_t split_half;
    int32_t activation;           /* DCTR_ACT_LINEAR | RELU | SIGMOID | TANH */
    int32_t workspace_ready;      /* 0: the call (re)writes `workspace` from filters[0] (one small launch in front of the kernel).
                                     1: the caller vouches `workspace` still holds what an earlier call with the same filters[0]
                                     VALUES, fields and layer_size[0] wrote: the fold launch is skipped (predict() over many batches) */
    const int32_t* layer_size;    /* HOST array [n_layers] */
    const float* const* filters;  /* HOST array of n_layers DEVICE pointers */
    const float* const* bias;     /* HOST array of n_layers DEVICE pointers */
    float* out;                   /* [B, featuremap_num] */
    void* workspace;              /* NULL, or device scratch of dctr_cin_workspace_bytes() bytes, 16-B aligned: layer 0 multiplies x_0
                                     with itself (z[i,j] = z[j,i]), so with a workspace the kernel walks the F0 (F0 + 1) / 2 pairs
                                     i <= j against W[ij] + W[ji] folded there (same result up to the rounding of that sum; 1.3x
                                     faster at C3).  Without it layer 0 walks all F0 x F0 products. */
    size_t workspace_bytes;
    float* const* save_y;         /* NULL, or HOST array of n_layers DEVICE pointers (entries may be NULL): layer k's
                                   * activations y_k [B*D, H_k] row-major (row b*D + d, ALL H_k maps) are also written there
                                   * — what dctr_cin_bwd otherwise recomputes with one GEMM per layer (ABI 4) */
} dctr_cin_args_t;
/* Bytes of `workspace` for these arguments (fields, dim, n_layers, layer_size, split_half are read; batch is not): the fold of layer 0,
 * plus — ABI 13 — for samples the kernel does not take whole, room for 1,024 samples of the sliced route.  CIN never mixes embedding
 * dimensions before its final reduce_sum over d (interaction.py:288-295, :322-323), and the reference puts no limit on embedding_dim:
 * a sample wider than one workgroup's MFMA tiles (embedding_dim > 128; > 64 when the maps of a 128-row tile would not fit the LDS) is
 * walked as dim / dd pseudo-samples of dd dimensions (dd = the largest divisor of dim the kernel takes) laid out in the workspace by a
 * pre-pass, their partial map sums added in the order of d by a post-pass; rows in chunks of what the workspace holds.  A network with
 * a layer of more maps than ANY tile height leaves LDS for (~480; the reference takes any layer_size) runs layer by layer as the
 * reference writes it (interaction.py:288-300): z = x_0 (outer) x_k materialised per chunk of samples, the 1 x 1 convolution on the
 * library's GEMM, bias + activation, the direct maps summed over d (room for up to 256 samples, at most 256 MiB unless 16 samples need
 * more).  For such arguments the workspace is REQUIRED (DCTR_E_NULL without, before anything is launched; any workspace with room for
 * >= 64 / >= 16 samples works); save_y is written as for whole samples (row b * dim + d). */
size_t dctr_cin_workspace_bytes(const dctr_cin_args_t* args);
int dctr_cin_fwd(const dctr_cin_args_t* args, void* stream);
/* ABI 13 — would dctr_cin_fwd (gather == NULL) / dctr_cin_gather_fwd (gather != NULL; fused_head != 0: with head_w / logit) take these
 * arguments?  Runs every shape check and kernel-shape decision of the call without launching; device pointers are not looked at
 * (args->layer_size, a HOST array, is; of `gather` the summary fields n_fields / batch / uniform_dim / all_dim4 / any_hash / any_identity /
 * any_pitch).  1 = yes, 0 = no (dctr_last_error() carries the reason).  Hosts ask this instead of re-deriving the kernels' limits. */
int dctr_cin_fwd_supported(const dctr_cin_args_t* args, const dctr_gather_fm_args_t* gather, int32_t fused_head);

/* ABI 8 — CIN.call over the embeddings of a gather, as deepctr/models/xdeepfm.py:52-66 wires it: exFM_in = concat_func(sparse_embedding_list,
 * axis=1) -> CIN -> Dense(1, use_bias=False).  The workgroup's [samples, F0, D] tile is read from the embedding tables inside the kernel
 * (gather: the dctr_embed_gather_fm arguments of the same batch — fields / ids / strides / status are used; every field a plain lookup
 * of width args->dim, dim % 4 == 0, no hashed and no pre-pooled field, else DCTR_E_UNSUPPORTED), args->x / x_stride are ignored and the
 * DNN input never has to exist in HBM.  head_w [featuremap_num] + logit [B] (both or neither): the Dense(1) over the summed maps is taken
 * on chip and only logit[b] leaves (args->out may then be NULL); without them args->out receives the maps as in dctr_cin_fwd.
 * An id outside its vocabulary reads row 0 and raises DCTR_STATUS_INDEX_OOR in *gather->status.  Inference only (save_y must be NULL). */
int dctr_cin_gather_fwd(const dctr_cin_args_t* args, const dctr_gather_fm_args_t* gather, const float* head_w, float* logit, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a11 AFMLayer.call — deepctr/layers/interaction.py:116-146 (inference: dropout inactive)
 *     x [B,F,E] (sample stride x_stride); W [E,A]; b [A]; h [A]; p [E]  ->  y [B]
 * ------------------------------------------------------------------------------------------------ */
int dctr_afm_fwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim, const float* att_w,
                 const float* att_b, const float* proj_h, const float* proj_p, int32_t att_factor, float* y,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * a12 InnerProductLayer.call — deepctr/layers/interaction.py:655-678
 *     x [B,F,E] (sample stride x_stride) -> y [B, F(F-1)/2] (reduce_sum) or [B, F(F-1)/2, E], sample stride y_stride;
 *     pair order (i<j) row-major.
 * ------------------------------------------------------------------------------------------------ */
int dctr_inner_product_fwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim,
                           int32_t reduce_sum, float* y, int64_t y_stride, void* stream);

/* ------------------------------------------------------------------------------------------------
 * sibling: CrossNetMix.call — deepctr/layers/interaction.py:511-549 (DCNMix)
 *     x [B, dim] (row stride x_stride) -> y [B, dim] (row stride y_stride)
 *     U, V [layers, experts, dim, low_rank], C [layers, experts, low_rank, low_rank] (U_list / V_list / C_list stacked
 *     over layers, :481-500), gating [experts, dim] (the experts' Dense(1, use_bias=False) kernels, shared by every
 *     layer, :502), bias [layers, dim].  layers == 0 copies x to y.  workspace: DEVICE scratch of
 *     dctr_crossnet_mix_workspace_bytes() bytes (U transposed per call so that the projection back reads it coalesced).
 * ------------------------------------------------------------------------------------------------ */
size_t dctr_crossnet_mix_workspace_bytes(int32_t dim, int32_t layers, int32_t experts, int32_t low_rank);
int dctr_crossnet_mix_fwd(const float* x, int64_t batch, int32_t dim, int64_t x_stride, const float* U, const float* V,
                          const float* C, const float* gating, const float* bias, int32_t layers, int32_t experts,
                          int32_t low_rank, float* y, int64_t y_stride, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * sibling: BiInteractionPooling.call — deepctr/layers/interaction.py:190-203 (NFM)
 *     x [B,F,E] (sample stride x_stride) -> y [B,E] (sample stride y_stride) = 0.5 ((sum_f e)^2 - sum_f e^2)
 * ------------------------------------------------------------------------------------------------ */
int dctr_bi_interaction_fwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim, float* y,
                            int64_t y_stride, void* stream);

/* ------------------------------------------------------------------------------------------------
 * adjacent  DNN.call (+ Dense(1, use_bias=False) head, add_func, PredictionLayer.call)
 *     deepctr/layers/core.py:189-208, :250-259; layers/utils.py:328-333.
 *     y = x; for each layer: y = act(y W_l + b_l).  Optional head: logit = y . head_w (+ add0 + add1
 *     + global_bias), sigmoid if binary.  W_l: [in_l, out_l] row-major (Keras kernel layout).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* x;               /* [B, x_stride] */
    int64_t batch;
    int64_t x_stride;
    int32_t in_dim;
    int32_t n_layers;
    const int32_t* units;         /* HOST array [n_layers] */
    const float* const* kernels;  /* HOST array of DEVICE pointers */
    const float* const* biases;   /* HOST array of DEVICE pointers */
    int32_t activation;           /* DCTR_ACT_* (DICE: dice_* arrays below) */
    int32_t has_head;
    const float* const* dice_alpha;   /* HOST arrays of DEVICE pointers [n_layers] or NULL */
    const float* const* dice_mean;
    const float* const* dice_var;
    float dice_eps;
    int32_t sigmoid_out;          /* PredictionLayer task == "binary" */
    const float* head_w;          /* [units[last]] or NULL */
    const float* add[4];          /* up to four [B] extra logit vectors (linear, FM, CIN ...) or NULL */
    const float* global_bias;     /* [1] device or NULL */
    float* y;                     /* has_head ? [B] : [B, y_stride] */
    int64_t y_stride;
    void* workspace;
    size_t workspace_bytes;
    float* const* save_acts;      /* training: HOST array [n_layers] of DEVICE pointers; layer l's activations
                                     [B, units[l]] (row stride units[l]) are also written there.  NULL = inference. */
    int32_t tile_rows;            /* batch rows per workgroup: 0 = auto, or 16 / 32 / 64.  Larger tiles re-use every
                                     weight fragment for more rows (less L2->CU weight traffic per row, the bound of
                                     this kernel) at the price of fewer workgroups; results are bit-identical.
                                     dctr_embed_mlp_fwd only: 64 = the streaming kernel (64-row tiles, LDS-DMA gather ring),
                                     128 / 256 = the row-chained kernel in one launch shape (waves own batch rows end to
                                     end; needs uniform_dim 16 / 32, a head, units as listed at uniform_dim; 128: units
                                     256-128-64 only).  0 (auto) sends launches of >= 64 rows per CU of such a model to the
                                     row-chained kernel as ONE launch: 256-row passes for the whole multiples of 256 rows x CUs,
                                     64-row units inside the same kernel for what is left; a row's result does not depend on
                                     the shape, the phase or the row's position in the launch. */
    int32_t precision;            /* must be 0: exact fp32 (v_mfma_f32_16x16x4_f32 = an fmaf chain) is the library's only arithmetic.  (ABI <= 11
                                     carried an exploratory bf16x3 mode here; removed — DESIGN.md §9 — anything but 0 is DCTR_E_UNSUPPORTED.) */
    unsigned long long* probe;    /* measurement aid, normally NULL: DEVICE uint64[2]; every workgroup does
                                     atomicMin(probe[0], t_start) / atomicMax(probe[1], t_end) with the constant-rate
                                     wall clock (dctr_wall_clock_khz()), so probe[1] - probe[0] is this launch's duration
                                     even inside a hipGraph, where event pairs cannot be attached.  Caller initialises
                                     to {UINT64_MAX, 0}. */
    const float* const* bn_scale; /* DNN(use_bn=True), inference form of keras BatchNormalization between bias_add and the
                                     activation (layers/core.py:200-201): HOST arrays [n_layers] of DEVICE pointers (entries or
                                     the arrays may be NULL = no BN on that layer): y = act((x W + b) * bn_scale + bn_shift) with
                                     bn_scale = gamma * rsqrt(moving_variance + epsilon), bn_shift = beta - moving_mean * bn_scale */
    const float* const* bn_shift;
    /* ABI 7 — CrossNet, vector parameterization (deepctr/layers/interaction.py:405-424), and the cross branch's share of DCN's
     * Dense(1) over Concatenate([cross_out, deep_out]) (deepctr/models/dcn.py:61-64), folded into the forward: with cross_layers = L
     * in 1 .. 3 the head also adds x_L . cross_head, where x_0 = the row the DNN reads (the gathered row of dctr_embed_mlp_fwd, or x)
     * and x_{l+1} = x_0 (x_l . cross_w[l]) + cross_b[l] + x_l.  Every x_l = a_l x_0 + (b_0 + .. + b_{l-1}), so the kernels take
     * L + 1 dot products of the row while it is on chip and run the scalar recurrence — the same real-number function as
     * dctr_crossnet_head_fwd, rounded differently (fp32 fmaf chains over the same products).  Needs has_head; not with save_acts or
     * precision != 0.  cross_w, cross_b: DEVICE [L, in_dim] row-major; cross_head: DEVICE [in_dim]. */
    const float* cross_w;
    const float* cross_b;
    const float* cross_head;
    int32_t cross_layers;         /* 0 (default): none */
    const float* cross_const;     /* NULL, or DEVICE float[4] written by dctr_crossnet_fold_consts() for THESE cross_w / cross_b /
                                   * cross_head: the L + 1 row-independent constants of the recurrence, which the kernels otherwise
                                   * compute at the start of every launch (one wave, ~7 dependent L2 round trips: a few us of a
                                   * 4096-row launch) */
} dctr_mlp_args_t;
/* Scratch dctr_mlp_fwd needs for these arguments (0 for most): a DNN with a layer wider
 * than any LDS tile holds (> 1,216 units — the reference's DNN takes any hidden_units, layers/core.py:160-175) -> two activation
 * buffers of [min(batch, 65536) rounded up to 64 rows, widest layer] floats: such a DNN runs layer by layer (own f32-MFMA GEMM +
 * one bias / BatchNormalization / activation launch per layer, rows in chunks of what the workspace holds; any workspace of >= 64
 * rows works), the head through the no-hidden-layer form of the same entry point. */
size_t dctr_mlp_workspace_bytes(const dctr_mlp_args_t* args);
/* Would dctr_mlp_fwd (g == NULL) / dctr_embed_mlp_fwd (g != NULL) take these arguments?  Runs every argument check and kernel-shape
 * decision of the launch without launching: 1 = yes, 0 = no (dctr_last_error() carries the reason).  Hosts ask this instead of
 * re-deriving the library's shape limits. */
int dctr_mlp_fwd_supported(const dctr_gather_fm_args_t* g, const dctr_mlp_args_t* m, int32_t add_fm_logit, int32_t add_lin_logit);
/* consts[l] = (cross_b[0] + .. + cross_b[l-1]) . v_l with v_l = cross_w[l] for l < layers and cross_head for l = layers (consts has
 * room for 4 floats; layers in 1 .. 3).  One tiny launch per change of the weights, not per batch. */
int dctr_crossnet_fold_consts(const float* cross_w, const float* cross_b, const float* cross_head, int32_t layers, int32_t dim,
                              float* consts, void* stream);
int dctr_mlp_fwd(const dctr_mlp_args_t* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a3-a8 + DNN fused: dctr_embed_gather_fm feeding dctr_mlp_fwd inside ONE launch.  The DNN-input tile is gathered
 * straight into LDS (g->dnn_in is ignored; m->x is ignored; m->in_dim must equal the gathered row width =
 * sum of the fields' dims at their out_offsets + dense_copy_cols), and the gather's FM / linear logits are added to
 * the head when the flags say so (they are also written to g->fm_logit / g->lin_logit when those are non-NULL).
 * Requirements: g->all_dim4, embedding_dim <= 64.  This is DeepFM's whole forward as a single kernel
 * (deepctr/models/deepfm.py:42-65).
 * ------------------------------------------------------------------------------------------------ */
int dctr_embed_mlp_fwd(const dctr_gather_fm_args_t* g, const dctr_mlp_args_t* m, int32_t add_fm_logit,
                       int32_t add_lin_logit, void* stream);

/* How dctr_embed_mlp_fwd would run this call.  Every path is ONE kernel launch; the row-chained kernel runs a call in up to two
 * phases inside that launch (256-row passes for the whole multiples of 256 rows x CUs, 64-row units for what is left).  Writes up
 * to `max` entries (rows, DCTR_FWD_KERNEL_* id, batch rows per workgroup) — one per phase, in order — and returns their number
 * (<= 0: error, see dctr_last_error).  Needs a current HIP device (CU count).
 * dctr_embed_mlp_fwd_last_kernel(): the DCTR_FWD_KERNEL_* id the calling thread's last successful dctr_embed_mlp_fwd call
 * launched (-1 before the first one).  DCTR_FWD_KERNEL_TILE covers both forms of the tile family: 16- / 32-row workgroups whose waves
 * stream their weight slices into registers, and — launches of at most 16 rows per CU (or tile_rows 16) with every layer width a
 * multiple of 16 — 16-row workgroups whose waves pull them by LDS-DMA (csrc/mlp_device.h: mlp_ring_kernel; same bits). */
enum { DCTR_FWD_KERNEL_TILE = 0, DCTR_FWD_KERNEL_STREAM = 1, DCTR_FWD_KERNEL_CHAIN = 2 };
int dctr_embed_mlp_fwd_plan(const dctr_gather_fm_args_t* g, const dctr_mlp_args_t* m, int64_t* rows, int32_t* kernel,
                            int32_t* rows_per_workgroup, int32_t max);
int dctr_embed_mlp_fwd_last_kernel(void);

/* ------------------------------------------------------------------------------------------------
 * a13 AttentionSequencePoolingLayer.call + LocalActivationUnit.call (DIN)
 *     deepctr/layers/sequence.py:261-298, layers/core.py:94-108, activation.py:59-64.
 *     query [B,E]; keys [B,T,E]; key_mask [B,T] bytes; att MLP over [q,k,q-k,q*k] (4E -> h1 -> .. -> 1)
 *     out [B,E] = sum_t score_t * k_t.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* query;
    const float* keys;
    const uint8_t* key_mask;
    int64_t batch;
    int32_t maxlen;
    int32_t dim;                  /* E (query / key width) */
    int32_t n_layers;
    int32_t activation;           /* DCTR_ACT_SIGMOID | RELU | DICE ... */
    const int32_t* units;         /* HOST */
    const float* const* kernels;  /* HOST array of DEVICE ptrs; kernels[0]: [4E, units[0]] */
    const float* const* biases;
    const float* const* dice_alpha;
    const float* const* dice_mean;
    const float* const* dice_var;
    float dice_eps;
    int32_t weight_normalization;
    const float* out_kernel;      /* [units[last]] */
    const float* out_bias;        /* [1] */
    float* out;                   /* [B, out_stride] */
    int64_t out_stride;
    float* scores;                /* optional [B,T] (return_score) */
    void* workspace;              /* optional device scratch of dctr_din_attn_workspace_bytes() bytes, 16-B aligned.
                                     With it (and dim % 16 == 0 and <= 64, layer widths <= 96, query / keys 16-B aligned) the
                                     attention MLP runs as one row problem over the B*T positions with the weights
                                     resident in LDS; without it, one workgroup per sample.  Layout (ABI 8): [B*T] raw scores,
                                     then the list of the positions that COUNT (masked positions are skipped by the row-chained
                                     score kernel: their scores never reach the output, sequence.py:280-285), then the
                                     workgroups' LDS image.  A workspace of only B*T floats (ABI <= 7 callers) still works: every
                                     position is scored and each workgroup folds the weights itself. */
    size_t workspace_bytes;
} dctr_din_attn_args_t;
size_t dctr_din_attn_workspace_bytes(const dctr_din_attn_args_t* args);
int dctr_din_attn_pool_fwd(const dctr_din_attn_args_t* args, void* stream);

/* ABI 7 — a13 with the lookups of the behaviour sequences and of the query features folded in (deepctr/models/sequence/din.py:62-76
 * embedding_lookup of query / keys + concat_func, layers/sequence.py:261-298): key row (b, t) = concat_h hist_table[h][hist_ids[h][b, t]],
 * query row b = concat_h query_table[h][query_ids[h][b]], read inside the score and pooling kernels — the [B, T, dim] key tensor and
 * the key mask never exist in HBM.  args->query / keys / key_mask are ignored; args->workspace ([B * T] floats) is required.
 * Position t of sample b counts iff hist_ids[h][b, t] != 0 for every feature with mask_zero[h] (keras Concat.compute_mask: the
 * conjunction of the mask_zero embeddings' masks); no mask_zero feature: every position counts.  An id outside its vocabulary reads
 * row 0 and raises DCTR_STATUS_INDEX_OOR in *status.  Plain ids only (hashed features: hash first, or use the lookups).
 * Covered: 1 or 2 features of equal width (dim / n_feats, a multiple of 16), dim in {16, 32, 64}, two-layer attention MLPs of the
 * row-chained score kernel; else DCTR_E_UNSUPPORTED (use dctr_embed_lookup_multi + dctr_din_attn_pool_fwd). */
typedef struct {
    int32_t n_feats;
    int32_t ids_is_i64;
    const void* hist_ids[2];      /* [B, maxlen] ids, row stride hist_stride (elements) */
    int64_t hist_stride;
    const void* query_ids[2];     /* [B] ids, element stride query_stride */
    int64_t query_stride;
    const float* hist_table[2];   /* [hist_vocab, dim / n_feats] fp32, 16-B aligned */
    const float* query_table[2];  /* [query_vocab, dim / n_feats] (usually the same table: shared embedding_name) */
    int64_t hist_vocab[2];
    int64_t query_vocab[2];
    int32_t mask_zero[2];
    int32_t* status;              /* DEVICE int or NULL */
} dctr_din_gather_t;
int dctr_din_attn_gather_fwd(const dctr_din_attn_args_t* args, const dctr_din_gather_t* gather, void* stream);

/* ================================================================================================
 * SURVEY.md §8(f) rank 1 — backward + optimizer of the hot path ("next" row; forward entry points above are
 * unchanged).  The reference has no code for these (Keras autodiff, tf.keras.optimizers.Adam): each entry is the
 * derivative of the forward expression of the cited lines.  Gradient tables are dense fp32 buffers of the same shape as
 * the parameter, ACCUMULATED into (atomics); dctr_adam_step consumes and clears them.
 * ================================================================================================ */

/* loss of PredictionLayer + compile(loss=...): task 0 = binary_crossentropy on sigmoid outputs (value with Keras'
 * 1e-7 clip), task 1 = mse.  dlogit[b] = d(mean loss)/d(logit_b); optional device floats: loss_sum += sum_b loss_b,
 * dlogit_sum += sum_b dlogit[b] (= gradient of PredictionLayer's global_bias, layers/core.py:250-259). */
int dctr_bce_grad(const float* pred, const float* y, int64_t batch, int32_t task, float* dlogit, float* loss_sum,
                  float* dlogit_sum, void* stream);
/* ABI 9 — the same with tf.keras' per-sample weights (Model.fit(sample_weight=, class_weight=), which the reference's models inherit):
 * loss = sum_b weight[b] * loss_b / batch (Keras' SUM_OVER_BATCH_SIZE reduction divides by the batch size, not by the summed weights),
 * so loss_sum += sum_b weight[b] loss_b and dlogit[b] = weight[b] * d(loss_b)/d(logit_b) / batch.  weight == NULL: dctr_bce_grad. */
int dctr_bce_grad_w(const float* pred, const float* y, const float* weight, int64_t batch, int32_t task, float* dlogit,
                    float* loss_sum, float* dlogit_sum, void* stream);

/* `touched` (ABI 6; optional, tables with dim % 4 == 0): one byte per 16-B group of g_table ([vocab * dim / 4] bytes, zero-initialised
 * by the caller).  The scatter kernels set the bytes of the groups they add to; dctr_opt_multi then treats groups with a clear byte as a
 * ZERO gradient it neither reads nor clears (see dctr_adam_seg_t).  Invariant the caller keeps: g_table is zero wherever its byte is. */
typedef struct {
    float* g_table;       /* [vocab, dim] gradient of the field's table, or NULL (frozen / absent)  */
    float* g_lin_table;   /* [vocab] gradient of its 1-wide linear table, or NULL                   */
    uint8_t* touched;     /* [vocab * dim / 4] bytes beside g_table, or NULL                        */
} dctr_field_grad_t;

/* backward of dctr_embed_gather_fm (inputs.py:101-117, layers/utils.py:336-346, feature_column.py:171-210,
 * layers/interaction.py:588-604):  d e_f = d_dnn_in[b, off_f..] + d_fm[b] * (sum_f' e_f' - e_f);  d lin_f[row] += d_lin[b];
 * d dense_lin_w[k] += d_lin[b] * dense[b,k]. */
typedef struct {
    const dctr_gather_fm_args_t* fwd;   /* the forward call's arguments (descriptors, ids, dense).  fwd->dnn_in, when not NULL, must
                                           still hold what that forward wrote in the fields' columns: the FM term reads e_f from
                                           there (coalesced) instead of the tables (a second random row read); NULL: from the tables */
    const dctr_field_grad_t* grads;     /* DEVICE array [n_fields], parallel to fwd->fields                  */
    const float* d_dnn_in;              /* [B, d_stride] gradient w.r.t. dnn_in, or NULL                      */
    int64_t d_stride;
    const float* d_fm;                  /* [B] gradient w.r.t. fm_logit, or NULL (model without FM)           */
    const float* d_lin;                 /* [B] gradient w.r.t. lin_logit, or NULL                             */
    float* g_dense_lin_w;               /* gradient of Linear.kernel, or NULL                                 */
    const int32_t* dense_lin_rows;      /* DEVICE [n_dense]: row of Linear.kernel fed by dense column k (-1: none);
                                           NULL = identity (fwd->dense_lin_w is the kernel itself)            */
} dctr_gather_fm_bwd_args_t;
int dctr_embed_gather_fm_bwd(const dctr_gather_fm_bwd_args_t* args, void* stream);

/* backward of dctr_embed_pool (inputs.py:120-158, layers/sequence.py:76-106, :155-183): the gradient of a pooled vector
 * (and of the pooled 1-wide linear term) scattered to the rows of the sequence with the forward's factors — weight *
 * mask (sum), / length (mean), or to the first position attaining each dimension's maximum (max). */
typedef struct {
    const dctr_pool_args_t* fwd;   /* the forward call's arguments                                      */
    const float* d_out;            /* [B, d_stride] gradient w.r.t. the pooled vectors, or NULL         */
    int64_t d_stride;
    const float* d_lin_out;        /* [B] gradient w.r.t. lin_out, or NULL                              */
    float* g_table;                /* [vocab, dim] accumulated, or NULL                                 */
    float* g_lin_table;            /* [vocab] accumulated, or NULL                                      */
    uint8_t* touched;              /* touched bytes of g_table (dctr_field_grad_t), or NULL (ABI 6)     */
} dctr_pool_bwd_args_t;
int dctr_embed_pool_bwd(const dctr_pool_bwd_args_t* args, void* stream);

/* backward of dctr_mlp_fwd with has_head (layers/core.py:189-208 + Dense(1, use_bias=False)): needs the activations the
 * forward saved through save_acts.  relu / linear / sigmoid / tanh: FIVE launches whatever the depth — head, W_l^T of every layer,
 * the backward chain (every dZ_l and dX in one launch on the forward's whole-MLP kernel), the dW_l (+ d_bias_l) row slices of every
 * layer as one grouped GEMM, the slice sum.  Dice: layer by layer (two GEMMs + the recomputed pre-activations per layer) on dctr_sgemm's
 * kernel. */
typedef struct {
    const float* x;               /* [B, x_stride] the forward's input (dnn_in)                          */
    int64_t batch;
    int64_t x_stride;
    int32_t in_dim;
    int32_t n_layers;             /* >= 1                                                                */
    const int32_t* units;         /* HOST [n_layers]                                                     */
    const float* const* kernels;  /* HOST array of DEVICE pointers, as in the forward                    */
    const float* const* acts;     /* HOST array of DEVICE pointers: saved activations [B, units[l]]      */
    int32_t activation;           /* DCTR_ACT_LINEAR | RELU | SIGMOID | TANH | DICE                      */
    int32_t pad_;
    const float* head_w;          /* [units[last]]                                                       */
    const float* dlogit;          /* [B]                                                                 */
    float* const* d_kernels;      /* HOST array of DEVICE pointers [in_l, out_l], accumulated into       */
    float* const* d_biases;       /* HOST array of DEVICE pointers [out_l] (or NULL entries), accumulated */
    float* d_head_w;              /* [units[last]], accumulated                                          */
    float* dx;                    /* [B, dx_stride] gradient w.r.t. x (written), or NULL                 */
    int64_t dx_stride;
    void* workspace;              /* dctr_mlp_bwd_workspace_bytes() bytes, 16-B aligned                  */
    size_t workspace_bytes;
    const float* d_out;           /* headless form (head_w == NULL): [B, d_out_stride] gradient w.r.t. the last
                                     layer's activations (the DNN branch of DCN feeds a wider Dense(1))  */
    int64_t d_out_stride;
    /* DCTR_ACT_DICE only (layers/activation.py:59-64 with the moving statistics): Dice is not invertible from its output,
     * so the pre-activations are recomputed (one more GEMM per layer) from `biases` and the layer inputs. */
    const float* const* biases;       /* HOST array of DEVICE pointers [out_l] (entries may be NULL), as in the forward */
    const float* const* dice_alpha;   /* HOST arrays of DEVICE pointers [out_l]                           */
    const float* const* dice_mean;
    const float* const* dice_var;
    float* const* d_dice_alpha;       /* accumulated; array or entries may be NULL                        */
    float dice_eps;
    int32_t pad2_;
    const float* const* dice_batch_mean;  /* Dice under training=True: HOST arrays of DEVICE pointers [out_l] with THIS batch's */
    const float* const* dice_batch_var;   /* statistics as dctr_dice_train_fwd returned them; the gradient then flows through
                                             them (BatchNormalization backward).  NULL (arrays or entries): dice_mean / dice_var
                                             are constants (inference statistics)                                            */
    void* dw_stream;              /* ABI 6.  NULL, or a second hipStream_t: the weight-gradient half of the chained form (the grouped
                                     dW / d_bias GEMM and its slice sum) is launched there behind an event on `stream`, so that it
                                     runs beside what the caller launches next on `stream` (the embedding scatter, atomics-bound).
                                     The CALLER joins the two streams (dctr_mlp_bwd_join) before it reads d_kernels / d_biases or
                                     reuses the workspace — and before anything on `stream` OVERWRITES what the side stream still
                                     reads: x, acts[l] (the layers' inputs X_l) and the dZ_l slices inside the workspace. */
    const float* const* saved_z;  /* ABI 6, DCTR_ACT_DICE.  NULL, or HOST array of DEVICE pointers [out_l] (entries may be NULL): layer l's
                                     pre-activations z_l = x_l W_l + b_l, dense [B, units[l]], as a forward launch wrote them (training-mode
                                     Dice runs the layers one by one and has them): the recompute GEMM of that layer is skipped         */
} dctr_mlp_bwd_args_t;
size_t dctr_mlp_bwd_workspace_bytes(const dctr_mlp_bwd_args_t* args);
/* `stream` waits (event record + wait, no host synchronisation) for everything issued to `dw_stream` so far.  No-op for NULL / equal streams. */
int dctr_mlp_bwd_join(void* stream, void* dw_stream);

/* Dice.call under training=True (deepctr/layers/activation.py:51-64: BatchNormalization(center=False, scale=False,
 * epsilon) with the training flag, then alpha (1 - p) z + p z):  z [rows, z_stride] pre-activations (+ bias[n] when bias !=
 * NULL); statistics = those of this batch over all rows (biased variance), written to batch_mean / batch_var [n]; the stored
 * statistics move towards them (moving = moving * momentum + batch * (1 - momentum); either may be NULL); h [rows, h_stride]
 * receives the activations.  The matching backward is dctr_mlp_bwd with dice_batch_mean / dice_batch_var. */
int dctr_dice_train_fwd(const float* z, int64_t z_stride, const float* bias, int64_t rows, int32_t n, const float* alpha,
                        float eps, float momentum, float* moving_mean, float* moving_var, float* batch_mean, float* batch_var,
                        float* h, int64_t h_stride, void* stream);
int dctr_mlp_bwd(const dctr_mlp_bwd_args_t* args, void* stream);

/* One DNN layer under training=True with the regularisers of the reference's DNN.call (deepctr/layers/core.py:196-208):
 *     fc = x W + b  ->  BatchNormalization(training) when use_bn  ->  activation  ->  Dropout(training) when dropout_rate > 0.
 * The dense part (z = x W + b) is a one-layer linear dctr_mlp_fwd; these two entries are what sits behind it and its backward:
 *   fwd:  [use_bn: batch mean / biased variance of z over all rows -> bn_batch_mean / bn_batch_var, stored statistics moved with
 *         bn_momentum (tf.keras: moving = moving * momentum + batch * (1 - momentum))]
 *         y = gamma (z - mean) rsqrt(var + eps) + beta   (y = z without BatchNormalization)
 *         h = keep(b, n) ? act(y) / (1 - rate) : 0        (keras' inverted dropout; keep = counter-based generator over
 *                                                          (dropout_seed, b * n + column): the backward regenerates it)
 *   bwd:  dy = dh * keep / (1 - rate) * act'(y);  d_beta += sum_b dy;  d_gamma += sum_b dy xhat;
 *         dz = gamma rsqrt(var + eps) (dy - mean_b(dy) - xhat mean_b(dy xhat))    (dz = dy without BatchNormalization)
 * The generator is this library's, not TensorFlow's: masks differ from a TF run with the same seed (as any two TF versions do). */
typedef struct {
    const float* z;               /* [rows, z_stride] pre-activations x W + b                              */
    int64_t z_stride;
    int64_t rows;
    int32_t n;                    /* columns (units of the layer)                                          */
    int32_t activation;           /* DCTR_ACT_LINEAR | RELU | SIGMOID | TANH                               */
    int32_t use_bn;
    float bn_eps;
    float bn_momentum;
    float dropout_rate;           /* in [0, 1); 0 = no dropout                                             */
    unsigned long long dropout_seed;
    const float* bn_gamma;        /* [n] or NULL (scale=False: 1)                                          */
    const float* bn_beta;         /* [n] or NULL (center=False: 0)                                         */
    float* bn_moving_mean;        /* [n] updated by the forward; NULL = leave                              */
    float* bn_moving_var;
    float* bn_batch_mean;         /* [n] written by the forward, read by the backward (use_bn)             */
    float* bn_batch_var;
    float* h;                     /* forward: [rows, h_stride] out                                         */
    int64_t h_stride;
    const float* dh;              /* backward: [rows, dh_stride] gradient w.r.t. h                         */
    int64_t dh_stride;
    float* dz;                    /* backward: [rows, n] contiguous out (may alias dh when dh_stride == n) */
    float* d_gamma;               /* backward: accumulated [n]; NULL ok                                    */
    float* d_beta;
    float* workspace;             /* backward with use_bn: 2 * n floats                                    */
} dctr_dnn_train_layer_t;
int dctr_dnn_train_layer_fwd(const dctr_dnn_train_layer_t* a, void* stream);
int dctr_dnn_train_layer_bwd(const dctr_dnn_train_layer_t* a, void* stream);

/* DIN's LocalActivationUnit as a training step (layers/core.py:94-108, layers/sequence.py:261-298, att_weight_normalization
 * = False): the attention input a[b*T+t, :] = [q_b, k_bt, q_b - k_bt, q_b * k_bt] ([B*T, 4*dim]) is materialised so that the
 * unit's MLP runs through dctr_mlp_fwd (save_acts) / dctr_mlp_bwd; out[b,:] = sum_t (mask ? score : 0) k[b,t,:].
 *   dctr_din_wsum_bwd:   d_score [B*T], dk [B,T,dim] WRITTEN (= masked score * d_out), d_bias += sum d_score (or NULL)
 *   dctr_din_att_in_bwd: da [B*T, 4*dim] -> dk ADDED to, dq added into dx[b, qcol[e]] (qcol: DEVICE int32 [dim], the columns of
 *                        the query embeddings inside the DNN input)
 *   dctr_embed_lookup_bwd: g_table[row(idx[i]), :] += d_out[i, :dim] with the forward's id resolution (hash, range check);
 *                        touched: the touched bytes of g_table (dctr_field_grad_t), or NULL (ABI 6) */
int dctr_din_att_in_fwd(const float* q, const float* k, int64_t batch, int32_t maxlen, int32_t dim, float* a, void* stream);
int dctr_din_wsum_fwd(const float* score, const uint8_t* mask, const float* k, int64_t batch, int32_t maxlen, int32_t dim,
                      float* out, int64_t out_stride, void* stream);
int dctr_din_wsum_bwd(const float* d_out, int64_t d_stride, const float* score, const uint8_t* mask, const float* k, int64_t batch,
                      int32_t maxlen, int32_t dim, float* d_score, float* dk, float* d_bias, void* stream);
/* att_weight_normalization=True (layers/sequence.py:283-289): p [B, T] = softmax over all T positions of where(mask, score, -2^32 + 1);
 * the weighted sum over the keys then takes p with an all-ones mask (dctr_din_wsum_fwd / _bwd).  Backward: d_score = mask ? p (dp - <p, dp>)
 * : 0 (d_score may alias dp), d_bias (NULL ok) += sum d_score. */
int dctr_din_softmax_fwd(const float* score, const uint8_t* mask, int64_t batch, int32_t maxlen, float* p, void* stream);
int dctr_din_softmax_bwd(const float* p, const uint8_t* mask, const float* dp, int64_t batch, int32_t maxlen, float* d_score, float* d_bias,
                         void* stream);
int dctr_din_att_in_bwd(const float* da, const float* q, const float* k, int64_t batch, int32_t maxlen, int32_t dim, float* dk,
                        float* dx, int64_t dx_stride, const int32_t* qcol, void* stream);
int dctr_embed_lookup_bwd(const dctr_lookup_args_t* fwd, const float* d_out, int64_t d_stride, float* g_table, uint8_t* touched,
                          void* stream);

/* backward of Dense(1, use_bias=False) on a strided [B, n] input: dx[b,:] = dlogit[b] * w (written), d_w += x^T dlogit. */
int dctr_dense1_bwd(const float* x, int64_t x_stride, int64_t batch, int32_t n, const float* w, const float* dlogit, float* dx,
                    int64_t dx_stride, float* d_w, void* stream);

/* backward of dctr_bi_interaction_fwd (interaction.py:190-203): dx[b,f,:] = dy[b,:] * (sum_f' x[b,f',:] - x[b,f,:]),
 * and of dctr_inner_product_fwd with reduce_sum (interaction.py:655-678): dx[b,i,:] = sum_{j != i} dy[b,pair(i,j)] x[b,j,:].
 * x as in the forward; dy [B, dy_stride]; dx [B, dx_stride] first F*E columns written, or added to with accumulate=1. */
int dctr_bi_interaction_bwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim, const float* dy,
                            int64_t dy_stride, float* dx, int64_t dx_stride, int32_t accumulate, void* stream);
int dctr_inner_product_bwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim, const float* dy,
                           int64_t dy_stride, float* dx, int64_t dx_stride, int32_t accumulate, void* stream);
/* backward of FM.call (interaction.py:588-604; dctr_fm_fwd / the gather's FM epilogue) for an FM group that is a slice of the DNN input
 * (DeepFM(fm_group=...) beyond the first group): dx[b,f,:] = dlogit[b] * (sum_f' x[b,f',:] - x[b,f,:]), written or added to. */
int dctr_fm_bwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim, const float* dlogit, float* dx,
                int64_t dx_stride, int32_t accumulate, void* stream);

/* backward of dctr_afm_fwd (AFMLayer.call, interaction.py:116-146); the attention is recomputed, nothing is saved by the
 * forward.  dy [B] = gradient w.r.t. the layer's [B,1] output; dx [B, dx_stride]: first fields*dim columns written (or
 * added to); d_att_w [dim, att_factor], d_att_b [att_factor], d_proj_h [att_factor], d_proj_p [dim] are ACCUMULATED. */
typedef struct {
    const float* x;               /* as in the forward: [B, x_stride], fields*dim used                    */
    int64_t batch;
    int64_t x_stride;
    int32_t fields, dim, att_factor, dx_accumulate;
    const float* att_w;
    const float* att_b;
    const float* proj_h;
    const float* proj_p;
    const float* dy;
    float* dx;
    int64_t dx_stride;
    float* d_att_w;
    float* d_att_b;
    float* d_proj_h;
    float* d_proj_p;
} dctr_afm_bwd_args_t;
int dctr_afm_bwd(const dctr_afm_bwd_args_t* args, void* stream);

/* backward of dctr_crossnet_fwd (interaction.py:405-424).  vector: one fused kernel (x_l recomputed) while a wave's x_l and its d w / d b
 * accumulators fit the LDS and a row's gradient the registers (dim <= 2048, 48 * layers * dim bytes <= 160 KiB); wider inputs (ABI 13: any
 * dim — Criteo at embedding_dim 64 is 1,677 columns) run the same recurrence layer by layer over the batch through the workspace.
 * matrix: GEMMs on dctr_sgemm's kernel + elementwise kernels through the workspace; u_l / x_l recomputed unless the forward saved
 * them (saved_u / saved_x).  dctr_crossnet_bwd_workspace_bytes() says what a call needs (0 for the on-chip vector form). */
typedef struct {
    const float* x;               /* [B, x_stride] the forward's input x_0                               */
    int64_t x_stride;
    int64_t batch;
    int32_t dim, layers, mode, dx_accumulate;   /* dx_accumulate: 1 = add into dx, 0 = overwrite          */
    const float* kernels;         /* as in the forward                                                   */
    const float* bias;
    const float* dy;              /* [B, dy_stride] gradient w.r.t. the output                           */
    int64_t dy_stride;
    float* d_kernels;             /* accumulated, same shape as kernels                                  */
    float* d_bias;                /* accumulated [layers, dim]                                           */
    float* dx;                    /* [B, dx_stride] gradient w.r.t. x_0                                  */
    int64_t dx_stride;
    void* workspace;              /* dctr_crossnet_bwd_workspace_bytes() bytes (matrix form; wide vector form), 16-B aligned */
    size_t workspace_bytes;
    const float* saved_u;         /* ABI 6, matrix form: NULL (u_l and x_l are recomputed: one GEMM + one elementwise launch per layer),   */
    const float* saved_x;         /* or what the forward wrote through dctr_crossnet_args_t.save_u / save_x (saved_x may be NULL: 1 layer) */
} dctr_crossnet_bwd_args_t;
size_t dctr_crossnet_bwd_workspace_bytes(const dctr_crossnet_bwd_args_t* args);
int dctr_crossnet_bwd(const dctr_crossnet_bwd_args_t* args, void* stream);

/* CrossNetMix backward (DCNMix; forward: dctr_crossnet_mix_fwd, deepctr/layers/interaction.py:511-549).  Nothing is saved by
 * the forward: x_l, the low-rank projections and the gate are recomputed into the workspace.  Weights as in the forward
 * (U, V [layers, experts, dim, low_rank], C [layers, experts, low_rank, low_rank], gating [experts, dim], bias [layers, dim]);
 * dU / dV / dC / dgating / dbias have the same shapes and are ACCUMULATED into; dx [B, dx_stride] = gradient w.r.t. x_0. */
typedef struct {
    const float* x;               /* [B, x_stride] the forward's input x_0                               */
    int64_t x_stride;
    int64_t batch;
    int32_t dim, layers, experts, low_rank;
    const float* U;
    const float* V;
    const float* C;
    const float* gating;
    const float* bias;
    const float* dy;              /* [B, dy_stride] gradient w.r.t. the output                           */
    int64_t dy_stride;
    float* dU;
    float* dV;
    float* dC;
    float* dgating;
    float* dbias;
    float* dx;                    /* [B, dx_stride]                                                      */
    int64_t dx_stride;
    int32_t dx_accumulate;        /* 1 = add into dx, 0 = overwrite                                      */
    int32_t pad_;
    void* workspace;              /* dctr_crossnet_mix_bwd_workspace_bytes() bytes, 16-B aligned         */
    size_t workspace_bytes;
} dctr_crossnet_mix_bwd_args_t;
size_t dctr_crossnet_mix_bwd_workspace_bytes(const dctr_crossnet_mix_bwd_args_t* args);
int dctr_crossnet_mix_bwd(const dctr_crossnet_mix_bwd_args_t* args, void* stream);

/* backward of dctr_cin_fwd (interaction.py:277-325).  Layers with H % 16 == 0, H <= 128, F0 <= 32, Fk <= 64 run z-free on two
 * MFMA kernels (filter gradient with x0*xk formed in registers; dz = dpre W^T contracted with x0 / xk tile by tile); other
 * shapes materialise z in the workspace and use dctr_sgemm's kernel.  The layer activations come from the forward call
 * (fwd->save_y -> saved_y) or are recomputed by re-running the forward kernel into the workspace. */
typedef struct {
    const dctr_cin_args_t* fwd;   /* the forward call's arguments (out / workspace unused)                */
    const float* d_out;           /* [B, out_dim] gradient w.r.t. the CIN output                          */
    int32_t out_dim;              /* featuremap_num                                                       */
    int32_t dx_accumulate;
    float* const* d_filters;      /* HOST array of DEVICE pointers, accumulated, shapes of fwd->filters   */
    float* const* d_bias;         /* HOST array of DEVICE pointers, accumulated                           */
    float* dx;                    /* [B, dx_stride] gradient w.r.t. x (first F0*D columns), or NULL       */
    int64_t dx_stride;
    void* workspace;              /* dctr_cin_bwd_workspace_bytes() bytes, 16-B aligned                   */
    size_t workspace_bytes;
    const float* const* saved_y;  /* NULL (the layer activations are recomputed), or HOST array of n_layers DEVICE
                                   * pointers to what the forward call wrote through fwd->save_y (ABI 4)   */
} dctr_cin_bwd_args_t;
size_t dctr_cin_bwd_workspace_bytes(const dctr_cin_bwd_args_t* args);
int dctr_cin_bwd(const dctr_cin_bwd_args_t* args, void* stream);

/* Keras Adam step over n contiguous floats (a whole table or weight): g' = g + 2*l2*w;  m = b1 m + (1-b1) g';
 * v = b2 v + (1-b2) g'^2;  w -= alpha * m / (sqrt(v) + eps) with alpha = lr*sqrt(1-b2^t)/(1-b1^t) from the caller.
 * Non-lazy like tf.keras' sparse apply: rows without a gradient still decay.  zero_grad clears g. */
int dctr_adam_step(float* w, float* m, float* v, float* g, int64_t n, float alpha, float beta1, float beta2, float eps,
                   float l2, int32_t zero_grad, void* stream);

/* the same for every parameter of a model in ONE launch: a DEVICE array of segments (16-B aligned buffers);
 * max_n = the largest segment's n (sizes the grid).  touched (ABI 6): NULL, or [n / 4] bytes — byte i clear means g[4i .. 4i+3] is
 * zero: the step does not read it (nor clear it); set bytes are cleared together with their gradient (zero_grad).  The update is the
 * same non-lazy one either way — every element of w / m / v moves. */
typedef struct {
    float* w;
    float* m;
    float* v;
    float* g;
    int64_t n;
    float l2;
    int32_t pad_;
    uint8_t* touched;
} dctr_adam_seg_t;
int dctr_adam_multi(const dctr_adam_seg_t* segs, int32_t n_segs, int64_t max_n, float alpha, float beta1, float beta2,
                    float eps, int32_t zero_grad, void* stream);

/* the other optimizers model.compile() accepts by name in the reference's examples, same segment array (tf.keras
 * formulas): ADAGRAD v += g^2, w -= lr g/(sqrt(v)+eps) (v starts at initial_accumulator_value, caller-filled);
 * RMSPROP (beta2 = rho) v = rho v + (1-rho) g^2, w -= lr g/(sqrt(v)+eps); SGD w -= lr g.  ADAM as dctr_adam_multi. */
enum { DCTR_OPT_ADAM = 0, DCTR_OPT_ADAGRAD = 1, DCTR_OPT_RMSPROP = 2, DCTR_OPT_SGD = 3 };
int dctr_opt_multi(int32_t kind, const dctr_adam_seg_t* segs, int32_t n_segs, int64_t max_n, float lr, float beta1,
                   float beta2, float eps, int32_t zero_grad, void* stream);
/* ABI 13 — the same step; with l2_penalty != NULL (a DEVICE double, accumulated into) the launch also adds
 *     penalty_scale * sum over the segments of l2 * sum(w^2),   taken on the weights BEFORE the update:
 * the regularisation losses tf.keras adds to that batch's reported loss (kernel_regularizer=l2(...) of inputs.py:22, layers/core.py:170,
 * interaction.py:100,258,387; tf.keras.Model.fit's `loss` is the batch-size-weighted mean of data loss + penalties: penalty_scale = the
 * batch's rows).  The weights pass through the step's registers anyway: no further pass over the tables. */
int dctr_opt_multi_l2(int32_t kind, const dctr_adam_seg_t* segs, int32_t n_segs, int64_t max_n, float lr, float beta1,
                      float beta2, float eps, int32_t zero_grad, double* l2_penalty, float penalty_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * the plain fp32 contractions of the backward step — dW = X^T dZ, dH = dZ W^T of DNN.call
 * (deepctr/layers/core.py:189-208 under Keras autodiff), the matrix CrossNet and CrossNetMix projections
 * (layers/interaction.py:405-424, :511-549) — on the library's own v_mfma_f32_16x16x4_f32 GEMM (csrc/gemm_kernels.hip; no BLAS
 * library behind this ABI).  Column-major BLAS semantics:  C (m x n, ldc) = op(A) (m x k) * op(B) (k x n) + beta * C, beta in {0, 1},
 * trans_x != 0: op = transpose; `batch` problems at the given element strides (batch 1: strides unused).  Exact fp32 products.
 * ------------------------------------------------------------------------------------------------ */
int dctr_sgemm(int32_t trans_a, int32_t trans_b, int32_t m, int32_t n, int32_t k, const float* A, int32_t lda, int64_t stride_a,
               const float* B, int32_t ldb, int64_t stride_b, float beta, float* C, int32_t ldc, int64_t stride_c, int32_t batch,
               void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCTR_H_ */

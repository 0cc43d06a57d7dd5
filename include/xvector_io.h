/* C-ABI of libxvector_io.so - the native Kaldi minibatch loader that feeds the MI355X x-vector engine.
 *
 * Host-only C++ (threads over memory-mapped arks), no HIP: it replaces the reference's pickle-over-multiprocessing loader
 *   dataset/data_loader.py:229-307 (batch_random), :310-414 (KaldiDataRandomQueue)
 *   dataset/kaldi_io.py:743-749, 814-867 (read_mat_from_segment / _read_compressed_submat: rows [start, start+T) of a
 *   Kaldi 'CM ' compressed matrix), :768-812 (the codec)
 * which the reference's own README names as its training bottleneck.  The caller hands in (pinned) host buffers;
 * a batch is  features [B][T][dim] float32  +  labels [B] int32  with B = num_speakers * num_segments and one T per
 * batch, exactly what Trainer.train feeds (model/trainer.py:491-508).
 *
 * Every function returns 0 on success; on failure a message is available from xvio_last_error() (thread-local).
 *
 * Arks are read through read-only shared mappings and must be IMMUTABLE while a loader that opened them is alive: truncating or
 * replacing a mapped ark (feature preparation re-run, NFS) ends the process with SIGBUS in a decoder thread.  XVIO_NO_MMAP=1 in the
 * environment keeps every read on pread(), which reports such a file as a clean "truncated" error instead.
 */
#ifndef XVECTOR_IO_H
#define XVECTOR_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xvio_loader xvio_loader;

typedef struct xvio_config {
    const char* data_dir;      /* Kaldi data directory: feats.scp, spk2utt, utt2num_frames (data_loader.py:14-54) */
    const char* spklist;       /* "<speaker> <int>" lines (train.py:74) */
    int32_t num_speakers;      /* speakers per batch */
    int32_t num_segments;      /* chunks per speaker */
    int32_t min_len, max_len;  /* one length T in [min_len, max_len] is drawn per batch (data_loader.py:273) */
    int32_t shuffle;           /* 1: random start frame, 0: frame 0 (kaldi_io.py:730-741) */
    int32_t num_threads;       /* decoder threads */
    int32_t queue_depth;       /* batches prepared ahead */
    uint64_t seed;             /* batch i is a pure function of (seed, i): same stream for any thread count */
    int32_t packed;            /* 1: the threads do not decode - a batch is delivered as the undecoded 'CM ' pieces of its rows
                                * (xvio_loader_next_packed) and decoded on the GPU (xv_cm_decode, xvector_hip.h): 1/4 of the host
                                * memory traffic and of the PCIe bytes.  'CM ' matrices only. */
} xvio_config;

const char* xvio_last_error(void);
int xvio_abi_version(void);

/* CRC32C (Castagnoli) of n bytes continuing from `crc` (0 to start): the checksum of TensorFlow V2 checkpoint files - the payload
 * tf.train.Saver writes at reference model/trainer.py:318,444 - which tf_kaldi_speaker_amd/misc/tf_checkpoint.py reads and writes. */
uint32_t xvio_crc32c(uint32_t crc, const void* data, uint64_t n);

/* Parses the directory (feats.scp / spk2utt / utt2num_frames / spklist), opens every ark once and starts the threads. */
int xvio_loader_create(const xvio_config* cfg, xvio_loader** out);
void xvio_loader_destroy(xvio_loader* l);

int xvio_loader_dim(const xvio_loader* l);               /* feature dimension */
int xvio_loader_total_speakers(const xvio_loader* l);    /* lines of spklist */
int xvio_loader_num_utterances(const xvio_loader* l);

/* Blocks until the next batch (in batch-index order) is ready and copies it out.
 * features: capacity >= B * max_len * dim floats, written as [B][*frames][dim]; labels: B ints. */
int xvio_loader_next(xvio_loader* l, float* features, int32_t* labels, int32_t* frames);

/* Packed mode.  One chunk = rows [start, start+frames) of one 'CM ' matrix (kaldi_io.py:814-867), undecoded:
 *   [min f32][range f32][dim x (p0, p25, p75, p100) u16][dim x frames u8, column after column], padded to a multiple of 16 bytes
 * = xvio_packed_chunk_bytes(dim, frames) bytes; a batch is B such chunks back to back.  packed: capacity >= B * chunk bytes at max_len. */
int64_t xvio_packed_chunk_bytes(int32_t dim, int32_t frames);
int xvio_loader_next_packed(xvio_loader* l, uint8_t* packed, int32_t* labels, int32_t* frames);

/* Batches decoded so far and the time the decoder threads spent on them (throughput reporting). */
int xvio_loader_stats(const xvio_loader* l, int64_t* batches, double* decode_seconds);

/* Codec entry point (also what the threads use): rows [start, start+length) of the matrix stored at byte `offset` of
 * `ark_path` (the "path:offset" of an scp line; the two bytes "\0B" are expected there).  'CM ' decodes only the
 * requested rows; 'FM ' / 'DM ' are sliced.  length < 0 reads everything from `start`.  out: capacity floats. */
int xvio_read_rows(const char* ark_path, int64_t offset, int32_t start, int32_t length, float* out, int64_t capacity,
                   int32_t* rows_out, int32_t* cols_out);

#ifdef __cplusplus
}
#endif
#endif

/* libasv_io.so - host-side I/O helper of asv-subtools_amd (no HIP in it; built by gcc next to libasv_amd.so).
 *
 * Replaces, on the sharded extraction path, the per-utterance read loop of the reference's extractor
 * (/root/reference/pytorch/pipeline/onestep/extract_embeddings.py:73-83: `for key, feats in kaldi_io.read_mat_scp(...)`,
 * one Python-level read + array per utterance): the payloads of a whole batch of scp entries are read straight into
 * their rows of one packed (pinned or pageable) batch buffer by a few native threads. */
#ifndef ASV_IO_H_
#define ASV_IO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASV_IO_VERSION 3
int asv_io_version(void);

/* n positioned reads: nbytes[i] bytes of descriptor fd[i] from file offset off[i] into dst[i], split over `threads` worker
 * threads (contiguous index ranges; <= 1: the calling thread).  Every read is completed (short reads are continued).
 * Returns 0, or -(i + 1) for the first read that failed or hit the end of its file early (errno-style detail in
 * asv_io_last_errno(): the value of the CALLING thread's last failing call - thread-local, so concurrent callers do not race). */
int asv_io_pread_batch(int n, const int32_t *fd, const int64_t *off, const int64_t *nbytes, void *const *dst, int threads);
int asv_io_last_errno(void);

/* Index of a Kaldi ark file of uncompressed float32 matrices - entries `key SP \0 B F M SP \4 <int32 rows> \4 <int32 cols> <payload>`,
 * what copy-feats / the reference's feature scripts write - scanned from byte `start` with small positioned reads (the payloads
 * are skipped).  Fills at most `cap` entries: payload_off[i], rows[i], cols[i], and the keys one after the other, '\n'-terminated,
 * into keys[0 .. keys_cap).  Returns the number of entries scanned; *next = file offset of the first entry NOT scanned;
 * *stopped = 0 end of file, 1 `cap` or `keys_cap` reached, 2 the entry at *next is of another kind (text, float64, compressed:
 * the caller falls back to its generic decoder), 3 malformed / truncated entry at *next, 4 read error (asv_io_last_errno()). */
int64_t asv_io_scan_ark(int fd, int64_t start, int64_t cap, int64_t *payload_off, int32_t *rows, int32_t *cols, char *keys, int64_t keys_cap,
                        int64_t *next, int32_t *stopped);

/* The binary ark entries `key SP \0 B F V SP \4 <int32 dim> <dim x float32>` of n row vectors (vectors[i * ld .. + dim)) in one buffer -
 * what the reference's extractor writes one kaldi_io.write_vec_flt() call at a time
 * (/root/reference/pytorch/pipeline/onestep/extract_embeddings.py:83, kaldi_io.py:367-399).  keys: the n keys, '\n'-separated (no
 * trailing separator needed, NUL-terminated).  Returns the bytes written, or -1 when out_cap is too small
 * (exact size: sum of key lengths + n * (11 + 4 * dim)). */
int64_t asv_io_pack_vec_ark(int n, int dim, const char *keys, const float *vectors, int64_t ld, char *out, int64_t out_cap);

/* (version 3) The text of a Kaldi scp table - one entry per non-blank line, `key WS rxfile`, what the reference reads one line at a time
 * (/root/reference/pytorch/libs/support/kaldi_io.py read_mat_scp) - parsed in one pass: for entry i (at most `cap`) the spans of its key and
 * rxfile in buf, and, when the rxfile has the plain form `path:digits`, path_id[i] = index of `path` among the distinct paths in order of first
 * appearance (their spans: path_off / path_len, at most path_cap; *n_paths of them) and offset[i] = the number; otherwise path_id[i] = -1
 * (range specifiers, pipes, bare files: the caller's generic reader).  Returns the number of entries, or -1 when cap / path_cap is too
 * small.  Why: with N ranks on one host every rank walks the WHOLE table (it needs every length to balance the shards) - in Python that
 * was 0.12 - 0.18 s per 50 000 entries and rank, the part of --sharded that did not scale (tools/bench_loaders.py). */
int64_t asv_io_parse_scp(const char *buf, int64_t len, int64_t cap, int64_t *key_off, int32_t *key_len, int64_t *rx_off, int32_t *rx_len,
                         int32_t *path_id, int64_t *offset, int32_t path_cap, int64_t *path_off, int32_t *path_len, int32_t *n_paths);

#ifdef __cplusplus
}
#endif
#endif

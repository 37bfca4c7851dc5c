/* libasv_io.so: batched positioned reads on native threads (include/asv_io.h).  Plain C, no HIP: the GPU library is not
 * touched by it. */
#define _GNU_SOURCE
#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>

#include "asv_io.h"

/* per calling thread: two loaders reading on two threads do not overwrite each other's detail.  The worker threads
 * of a batch hand their errno back through their range_t; it is published here by the thread that made the call. */
static __thread int last_errno = 0;

int asv_io_version(void) { return ASV_IO_VERSION; }
int asv_io_last_errno(void) { return last_errno; }

typedef struct {
  int lo, hi;
  const int32_t *fd;
  const int64_t *off, *nbytes;
  void *const *dst;
  int failed;          /* index + 1 of the first failing read of this range, 0 = none */
  int err;
} range_t;

static void *run_range(void *arg) {
  range_t *r = (range_t *)arg;
  for (int i = r->lo; i < r->hi; ++i) {
    int64_t got = 0;
    char *p = (char *)r->dst[i];
    while (got < r->nbytes[i]) {
      const ssize_t k = pread(r->fd[i], p + got, (size_t)(r->nbytes[i] - got), (off_t)(r->off[i] + got));
      if (k < 0 && errno == EINTR) continue;
      if (k <= 0) {                                   /* error, or the file ends inside the matrix */
        r->failed = i + 1;
        r->err = k < 0 ? errno : 0;
        return NULL;
      }
      got += k;
    }
  }
  return NULL;
}

int asv_io_pread_batch(int n, const int32_t *fd, const int64_t *off, const int64_t *nbytes, void *const *dst, int threads) {
  if (n <= 0) return 0;
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  if (threads > n) threads = n;
  range_t r[64];
  pthread_t th[64];
  const int step = (n + threads - 1) / threads;
  int used = 0;
  for (int t = 0; t < threads; ++t) {
    const int lo = t * step, hi = lo + step < n ? lo + step : n;
    if (lo >= hi) break;
    r[used] = (range_t){lo, hi, fd, off, nbytes, dst, 0, 0};
    ++used;
  }
  int started = 0;
  for (int t = 1; t < used; ++t) {                    /* range 0 runs on the calling thread */
    if (pthread_create(&th[t], NULL, run_range, &r[t]) != 0) break;
    started = t;
  }
  run_range(&r[0]);
  for (int t = started + 1; t < used; ++t) run_range(&r[t]);      /* threads that could not be created: do their work here */
  for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
  for (int t = 0; t < used; ++t)
    if (r[t].failed) {
      last_errno = r[t].err;
      return -r[t].failed;
    }
  return 0;
}

int64_t asv_io_scan_ark(int fd, int64_t start, int64_t cap, int64_t *payload_off, int32_t *rows, int32_t *cols, char *keys, int64_t keys_cap,
                        int64_t *next, int32_t *stopped) {
  enum { CHUNK = 512 };
  unsigned char buf[CHUNK];
  int64_t pos = start, n = 0, kused = 0;
  *stopped = 0;
  while (n < cap) {
    ssize_t got;
    do { got = pread(fd, buf, CHUNK, (off_t)pos); } while (got < 0 && errno == EINTR);
    if (got < 0) { last_errno = errno; *stopped = 4; break; }
    /* skip what Kaldi allows between entries: nothing in binary archives, but tolerate trailing whitespace at the end of the file */
    ssize_t b = 0;
    while (b < got && (buf[b] == '\n' || buf[b] == ' ' || buf[b] == '\t' || buf[b] == '\r')) ++b;
    if (b == got) {
      if (got < CHUNK) break;                          /* end of file */
      pos += b;
      continue;
    }
    ssize_t sp = b;
    while (sp < got && buf[sp] != ' ') ++sp;
    if (sp >= got || sp == b) { *stopped = 3; pos += b; break; }       /* no key terminator within the chunk (keys are short) / empty key */
    const ssize_t klen = sp - b;
    if (sp + 1 + 15 > got) { *stopped = (sp + 1 + 2 <= got && !(buf[sp + 1] == 0 && buf[sp + 2] == 'B')) ? 2 : 3; pos += b; break; }
    const unsigned char *h = buf + sp + 1;
    if (!(h[0] == 0 && h[1] == 'B' && h[2] == 'F' && h[3] == 'M' && h[4] == ' ' && h[5] == 4 && h[10] == 4)) { *stopped = 2; pos += b; break; }
    int32_t r, c;
    __builtin_memcpy(&r, h + 6, 4);
    __builtin_memcpy(&c, h + 11, 4);
    if (r < 0 || c < 0) { *stopped = 3; pos += b; break; }
    if (kused + klen + 1 > keys_cap) { *stopped = 1; pos += b; break; }
    __builtin_memcpy(keys + kused, buf + b, (size_t)klen);
    keys[kused + klen] = '\n';
    kused += klen + 1;
    payload_off[n] = pos + sp + 1 + 15;
    rows[n] = r;
    cols[n] = c;
    pos = payload_off[n] + (int64_t)r * c * 4;
    ++n;
    if (n == cap) { *stopped = 1; break; }
  }
  *next = pos;
  return n;
}

int64_t asv_io_pack_vec_ark(int n, int dim, const char *keys, const float *vectors, int64_t ld, char *out, int64_t out_cap) {
  /* key SP \0 B F V SP \4 <int32 dim> <dim x f32> per vector: the bytes kaldi_io.write_vec_flt writes one entry at a time */
  const char *k = keys;
  int64_t used = 0;
  for (int i = 0; i < n; ++i) {
    const char *e = k;
    while (*e != '\n' && *e != 0) ++e;
    const int64_t klen = e - k, need = klen + 1 + 10 + (int64_t)dim * 4;
    if (used + need > out_cap) return -1;
    memcpy(out + used, k, (size_t)klen);
    used += klen;
    out[used++] = ' ';
    out[used++] = 0; out[used++] = 'B'; out[used++] = 'F'; out[used++] = 'V'; out[used++] = ' '; out[used++] = 4;
    const int32_t d = dim;
    memcpy(out + used, &d, 4);
    used += 4;
    memcpy(out + used, vectors + (int64_t)i * ld, (size_t)dim * 4);
    used += (int64_t)dim * 4;
    k = (*e == '\n') ? e + 1 : e;
  }
  return used;
}

int64_t asv_io_parse_scp(const char *buf, int64_t len, int64_t cap, int64_t *key_off, int32_t *key_len, int64_t *rx_off, int32_t *rx_len,
                         int32_t *path_id, int64_t *offset, int32_t path_cap, int64_t *path_off, int32_t *path_len, int32_t *n_paths) {
  int64_t n = 0, pos = 0;
  int32_t np = 0, last = -1;
  while (pos < len) {
    int64_t eol = pos;
    while (eol < len && buf[eol] != '\n') ++eol;
    int64_t a = pos, b = eol;                            /* the line without surrounding white space */
    while (a < b && (buf[a] == ' ' || buf[a] == '\t' || buf[a] == '\r')) ++a;
    while (b > a && (buf[b - 1] == ' ' || buf[b - 1] == '\t' || buf[b - 1] == '\r')) --b;
    pos = eol + 1;
    if (a == b) continue;                                /* blank line */
    if (n == cap) return -1;
    int64_t ke = a;
    while (ke < b && buf[ke] != ' ' && buf[ke] != '\t') ++ke;
    int64_t rs = ke;
    while (rs < b && (buf[rs] == ' ' || buf[rs] == '\t')) ++rs;
    key_off[n] = a; key_len[n] = (int32_t)(ke - a);
    rx_off[n] = rs; rx_len[n] = (int32_t)(b - rs);
    path_id[n] = -1; offset[n] = 0;
    /* plain form: path ':' digits (range specifiers end in ']', pipes in '|': neither ends in a digit run behind a colon) */
    int64_t d = b;
    while (d > rs && buf[d - 1] >= '0' && buf[d - 1] <= '9') --d;
    if (d < b && d > rs + 1 && buf[d - 1] == ':' && b - d <= 18) {
      int64_t v = 0;
      for (int64_t q = d; q < b; ++q) v = v * 10 + (buf[q] - '0');
      const int64_t po = rs;
      const int32_t pl = (int32_t)(d - 1 - rs);
      int32_t id = -1;
      if (last >= 0 && path_len[last] == pl && memcmp(buf + path_off[last], buf + po, (size_t)pl) == 0) id = last;
      else {
        for (int32_t k = 0; k < np; ++k)
          if (path_len[k] == pl && memcmp(buf + path_off[k], buf + po, (size_t)pl) == 0) { id = k; break; }
        if (id < 0) {
          if (np == path_cap) return -1;
          path_off[np] = po; path_len[np] = pl;
          id = np++;
        }
      }
      last = id;
      path_id[n] = id; offset[n] = v;
    }
    ++n;
  }
  *n_paths = np;
  return n;
}

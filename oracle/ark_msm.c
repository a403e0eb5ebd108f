/* TEST INFRASTRUCTURE ONLY -- plain-C CPU restatement of the MSM hot path ("ark-ec-equivalent").
 *
 * PARITY UNPINNED.  The algorithm lives in third-party crates that are absent from /root/reference
 * and not pinned by it: ark-ec / ark-ff `^0.2.0` (Cargo.toml:15-16, no Cargo.lock: .gitignore:2),
 * ark-poly-commit @ git branch `accumulation-experimental` (Cargo.toml:34).  The reference's own tests
 * contain no golden vectors (src/lib.rs:334-395).  This file restates the published algorithm of
 * ark-ec 0.2 `VariableBaseMSM::multi_scalar_mul` (structure as recalled in SURVEY.md Appendix C):
 *   c = 3 if n < 32 else ceil_log2(n)*69/100 + 2;   windows at 0, c, 2c, ... < MODULUS_BITS;
 *   per window: 2^c - 1 Jacobian buckets, zero scalars skipped, scalar == 1 added once in window 0,
 *   mixed addition per pair, running-sum bucket reduction, Horner combine with c doublings;
 *   windows are independent (rayon `parallel` feature => one thread per window here).
 * with Jacobian formulas madd-2007-bl / add-2007-bl / dbl-2009-l (a = 0), 64-bit-limb Montgomery
 * fields (4 limbs Pallas + BLS12-381 Fr, 6 limbs BLS12-381 Fq).  It is validated against the
 * independent big-integer oracle oracle/pyref.py (tests/test_oracle.py) and the committed fixtures in
 * tests/golden/.  Results are canonical affine points, so they do not depend on the algorithm.
 *
 * Uses: (1) large-N checker for the GPU path in tests/, (2) bench.py's `cpu_baseline` leg
 * (kind "port").  Nothing in the product (accumulation_amd/, libamsm.so) links or loads this.
 *
 * Reference call sites this stands in for (paths under /root/reference):
 *   PedersenCommitment::commit -> src/hp_as/mod.rs:196,197,214,377,911-918;
 *   src/r1cs_nark_as/mod.rs:394-410,1081-1093; src/r1cs_nark_as/r1cs_nark/mod.rs:216-261,375-403;
 *   compute_hp src/hp_as/mod.rs:278-285; combine_vectors :492-512.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define MAXL 6

typedef struct {
  int n;        /* limbs */
  u64 m[MAXL];  /* modulus */
  u64 one[MAXL];
  u64 r2[MAXL];
  u64 inv; /* -m^-1 mod 2^64 */
} field_t;

typedef struct {
  field_t fq; /* coordinates */
  field_t fr; /* scalars */
  int scalar_bits;
  u64 gx[MAXL], gy[MAXL]; /* generator, canonical */
} curve_t;

static u64 inv64(u64 m) {
  u64 x = 1;
  for (int i = 0; i < 6; i++) x *= 2 - m * x;
  return (u64)0 - x;
}

static int geq(const u64* a, const u64* b, int n) {
  for (int i = n - 1; i >= 0; i--) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
static u64 sub_n(u64* r, const u64* a, const u64* b, int n) {
  u64 br = 0;
  for (int i = 0; i < n; i++) {
    u128 x = (u128)a[i] - b[i] - br;
    r[i] = (u64)x;
    br = (u64)(x >> 64) & 1;
  }
  return br;
}
static u64 add_n(u64* r, const u64* a, const u64* b, int n) {
  u128 c = 0;
  for (int i = 0; i < n; i++) {
    c += (u128)a[i] + b[i];
    r[i] = (u64)c;
    c >>= 64;
  }
  return (u64)c;
}
static int is_zero(const u64* a, int n) {
  u64 o = 0;
  for (int i = 0; i < n; i++) o |= a[i];
  return o == 0;
}
static int eq_n(const u64* a, const u64* b, int n) { return memcmp(a, b, 8 * n) == 0; }

static void f_add(const field_t* f, u64* r, const u64* a, const u64* b) {
  u64 c = add_n(r, a, b, f->n);
  if (c || geq(r, f->m, f->n)) sub_n(r, r, f->m, f->n);
}
static void f_sub(const field_t* f, u64* r, const u64* a, const u64* b) {
  if (sub_n(r, a, b, f->n)) add_n(r, r, f->m, f->n);
}
static void f_dbl(const field_t* f, u64* r, const u64* a) { f_add(f, r, a, a); }
static void f_neg(const field_t* f, u64* r, const u64* a) {
  if (is_zero(a, f->n)) {
    memcpy(r, a, 8 * f->n);
    return;
  }
  sub_n(r, f->m, a, f->n);
}
static void f_mul(const field_t* f, u64* r, const u64* a, const u64* b) {
  int n = f->n;
  u64 t[MAXL + 2];
  memset(t, 0, sizeof(t));
  for (int i = 0; i < n; i++) {
    u128 c = 0;
    for (int j = 0; j < n; j++) {
      c += (u128)a[j] * b[i] + t[j];
      t[j] = (u64)c;
      c >>= 64;
    }
    c += t[n];
    t[n] = (u64)c;
    t[n + 1] = (u64)(c >> 64);
    u64 m = t[0] * f->inv;
    c = ((u128)m * f->m[0] + t[0]) >> 64;
    for (int j = 1; j < n; j++) {
      c += (u128)m * f->m[j] + t[j];
      t[j - 1] = (u64)c;
      c >>= 64;
    }
    c += t[n];
    t[n - 1] = (u64)c;
    t[n] = t[n + 1] + (u64)(c >> 64);
  }
  if (t[n] || geq(t, f->m, n)) sub_n(t, t, f->m, n);
  memcpy(r, t, 8 * n);
}
static void f_sqr(const field_t* f, u64* r, const u64* a) { f_mul(f, r, a, a); }
static void f_inv(const field_t* f, u64* r, const u64* a) {
  u64 e[MAXL], two[MAXL] = {2, 0, 0, 0, 0, 0}, acc[MAXL], base[MAXL];
  sub_n(e, f->m, two, f->n);
  memcpy(acc, f->one, 8 * f->n);
  memcpy(base, a, 8 * f->n);
  for (int i = f->n * 64 - 1; i >= 0; i--) {
    f_sqr(f, acc, acc);
    if ((e[i >> 6] >> (i & 63)) & 1) f_mul(f, acc, acc, base);
  }
  memcpy(r, acc, 8 * f->n);
}
static void f_to_mont(const field_t* f, u64* r, const u64* a) { f_mul(f, r, a, f->r2); }
static void f_from_mont(const field_t* f, u64* r, const u64* a) {
  u64 o[MAXL] = {1, 0, 0, 0, 0, 0};
  f_mul(f, r, a, o);
}

static void field_init(field_t* f, int n, const u64* m) {
  f->n = n;
  memcpy(f->m, m, 8 * n);
  f->inv = inv64(m[0]);
  /* R mod m and R^2 mod m by repeated doubling of 1 (2*64n doublings) */
  u64 x[MAXL] = {1, 0, 0, 0, 0, 0};
  for (int i = 0; i < 64 * n; i++) {
    u64 c = add_n(x, x, x, n);
    if (c || geq(x, m, n)) sub_n(x, x, m, n);
  }
  memcpy(f->one, x, 8 * n);
  for (int i = 0; i < 64 * n; i++) {
    u64 c = add_n(x, x, x, n);
    if (c || geq(x, m, n)) sub_n(x, x, m, n);
  }
  memcpy(f->r2, x, 8 * n);
}

static curve_t g_curves[2];
static int g_init = 0;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

static void curves_init(void) {
  pthread_mutex_lock(&g_lock);
  if (!g_init) {
    static const u64 pallas_p[4] = {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0ull, 0x4000000000000000ull};
    static const u64 pallas_r[4] = {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0ull, 0x4000000000000000ull};
    static const u64 bls_p[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    static const u64 bls_r[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull,
                                 0x73eda753299d7d48ull};
    static const u64 bls_gx[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull,
                                  0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
    static const u64 bls_gy[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull,
                                  0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};
    curve_t* c = &g_curves[0];
    memset(g_curves, 0, sizeof(g_curves));
    field_init(&c->fq, 4, pallas_p);
    field_init(&c->fr, 4, pallas_r);
    c->scalar_bits = 255;
    /* generator (-1, 2) */
    u64 one[MAXL] = {1, 0, 0, 0, 0, 0};
    sub_n(c->gx, pallas_p, one, 4);
    c->gy[0] = 2;
    c = &g_curves[1];
    field_init(&c->fq, 6, bls_p);
    field_init(&c->fr, 4, bls_r);
    c->scalar_bits = 255;
    memcpy(c->gx, bls_gx, 48);
    memcpy(c->gy, bls_gy, 48);
    g_init = 1;
  }
  pthread_mutex_unlock(&g_lock);
}

/* ---- Jacobian points (ark-ec GroupProjective), a = 0 ------------------------------------------ */
typedef struct {
  u64 x[MAXL], y[MAXL], z[MAXL];
} jac_t;

static void jac_set_inf(const field_t* f, jac_t* p) {
  memset(p, 0, sizeof(*p));
  memcpy(p->x, f->one, 8 * f->n);
  memcpy(p->y, f->one, 8 * f->n);
}
static int jac_is_inf(const field_t* f, const jac_t* p) { return is_zero(p->z, f->n); }

/* dbl-2009-l */
static void jac_dbl(const field_t* f, jac_t* p) {
  if (jac_is_inf(f, p)) return;
  u64 a[MAXL], b[MAXL], c[MAXL], d[MAXL], e[MAXL], ff[MAXL], t[MAXL];
  f_sqr(f, a, p->x);
  f_sqr(f, b, p->y);
  f_sqr(f, c, b);
  f_add(f, d, p->x, b);
  f_sqr(f, d, d);
  f_sub(f, d, d, a);
  f_sub(f, d, d, c);
  f_dbl(f, d, d);
  f_dbl(f, e, a);
  f_add(f, e, e, a);
  f_sqr(f, ff, e);
  f_mul(f, p->z, p->y, p->z);
  f_dbl(f, p->z, p->z);
  f_sub(f, p->x, ff, d);
  f_sub(f, p->x, p->x, d);
  f_sub(f, t, d, p->x);
  f_mul(f, t, e, t);
  f_dbl(f, c, c);
  f_dbl(f, c, c);
  f_dbl(f, c, c);
  f_sub(f, p->y, t, c);
}

/* madd-2007-bl: p += (qx, qy) affine */
static void jac_madd(const field_t* f, jac_t* p, const u64* qx, const u64* qy) {
  if (is_zero(qx, f->n) && is_zero(qy, f->n)) return; /* affine infinity encoded (0,0) */
  if (jac_is_inf(f, p)) {
    memcpy(p->x, qx, 8 * f->n);
    memcpy(p->y, qy, 8 * f->n);
    memcpy(p->z, f->one, 8 * f->n);
    return;
  }
  u64 z1z1[MAXL], u2[MAXL], s2[MAXL], h[MAXL], hh[MAXL], i[MAXL], j[MAXL], r[MAXL], v[MAXL], t[MAXL];
  f_sqr(f, z1z1, p->z);
  f_mul(f, u2, qx, z1z1);
  f_mul(f, s2, qy, p->z);
  f_mul(f, s2, s2, z1z1);
  if (eq_n(p->x, u2, f->n)) {
    if (eq_n(p->y, s2, f->n)) {
      jac_dbl(f, p);
    } else {
      jac_set_inf(f, p);
    }
    return;
  }
  f_sub(f, h, u2, p->x);
  f_sqr(f, hh, h);
  f_dbl(f, i, hh);
  f_dbl(f, i, i);
  f_mul(f, j, h, i);
  f_sub(f, r, s2, p->y);
  f_dbl(f, r, r);
  f_mul(f, v, p->x, i);
  f_add(f, t, p->z, h); /* Z3 = (Z1+H)^2 - Z1Z1 - HH */
  f_sqr(f, t, t);
  f_sub(f, t, t, z1z1);
  f_sub(f, p->z, t, hh);
  f_sqr(f, p->x, r);
  f_sub(f, p->x, p->x, j);
  f_sub(f, p->x, p->x, v);
  f_sub(f, p->x, p->x, v);
  f_sub(f, t, v, p->x);
  f_mul(f, t, r, t);
  f_mul(f, j, p->y, j);
  f_dbl(f, j, j);
  f_sub(f, p->y, t, j);
}

/* add-2007-bl */
static void jac_add(const field_t* f, jac_t* p, const jac_t* q) {
  if (jac_is_inf(f, q)) return;
  if (jac_is_inf(f, p)) {
    *p = *q;
    return;
  }
  u64 z1z1[MAXL], z2z2[MAXL], u1[MAXL], u2[MAXL], s1[MAXL], s2[MAXL], h[MAXL], i[MAXL], j[MAXL], r[MAXL], v[MAXL],
      t[MAXL];
  f_sqr(f, z1z1, p->z);
  f_sqr(f, z2z2, q->z);
  f_mul(f, u1, p->x, z2z2);
  f_mul(f, u2, q->x, z1z1);
  f_mul(f, s1, p->y, q->z);
  f_mul(f, s1, s1, z2z2);
  f_mul(f, s2, q->y, p->z);
  f_mul(f, s2, s2, z1z1);
  if (eq_n(u1, u2, f->n)) {
    if (eq_n(s1, s2, f->n)) {
      jac_dbl(f, p);
    } else {
      jac_set_inf(f, p);
    }
    return;
  }
  f_sub(f, h, u2, u1);
  f_dbl(f, i, h);
  f_sqr(f, i, i);
  f_mul(f, j, h, i);
  f_sub(f, r, s2, s1);
  f_dbl(f, r, r);
  f_mul(f, v, u1, i);
  f_add(f, t, p->z, q->z);
  f_sqr(f, t, t);
  f_sub(f, t, t, z1z1);
  f_sub(f, t, t, z2z2);
  f_mul(f, p->z, t, h);
  f_sqr(f, p->x, r);
  f_sub(f, p->x, p->x, j);
  f_sub(f, p->x, p->x, v);
  f_sub(f, p->x, p->x, v);
  f_sub(f, t, v, p->x);
  f_mul(f, t, r, t);
  f_mul(f, s1, s1, j);
  f_dbl(f, s1, s1);
  f_sub(f, p->y, t, s1);
}

static void jac_to_affine(const field_t* f, const jac_t* p, u64* xy, uint8_t* inf) {
  if (jac_is_inf(f, p)) {
    memset(xy, 0, 16 * f->n);
    *inf = 1;
    return;
  }
  u64 zi[MAXL], zi2[MAXL];
  f_inv(f, zi, p->z);
  f_sqr(f, zi2, zi);
  f_mul(f, xy, p->x, zi2);
  f_mul(f, zi2, zi2, zi);
  f_mul(f, xy + f->n, p->y, zi2);
  *inf = 0;
}

/* ---- ark-ec 0.2 style Pippenger ---------------------------------------------------------------- */
static int ln_without_floats(size_t a) { /* ceil_log2(a) * 69 / 100 */
  int l = 0;
  while (((size_t)1 << l) < a) l++;
  return l * 69 / 100;
}

typedef struct {
  const curve_t* cv;
  const u64* bases; /* n * 2L, Montgomery affine, (0,0) = infinity */
  const u64* scalars; /* n * 4 canonical */
  size_t n;
  int c;
  int n_windows;
  jac_t* window_sums;
  volatile int next; /* work queue over windows */
  pthread_mutex_t lock;
} msm_job_t;

static u64 scalar_digit(const u64* s, int w_start, int c) {
  /* (scalar >> w_start) mod 2^c over a 256-bit little-endian integer */
  int limb = w_start >> 6, sh = w_start & 63;
  u64 v = s[limb] >> sh;
  if (sh && limb + 1 < 4) v |= s[limb + 1] << (64 - sh);
  return c >= 64 ? v : (v & (((u64)1 << c) - 1));
}

static void msm_window(msm_job_t* job, int wi) {
  const field_t* f = &job->cv->fq;
  int L = f->n;
  int c = job->c;
  int w_start = wi * c;
  size_t nb = ((size_t)1 << c) - 1;
  jac_t* buckets = (jac_t*)malloc(nb * sizeof(jac_t));
  jac_t res;
  jac_set_inf(f, &res);
  for (size_t b = 0; b < nb; b++) jac_set_inf(f, &buckets[b]);
  for (size_t i = 0; i < job->n; i++) {
    const u64* s = job->scalars + 4 * i;
    if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;
    const u64* bx = job->bases + 2 * L * i;
    if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) {
      if (w_start == 0) jac_madd(f, &res, bx, bx + L);
    } else {
      u64 d = scalar_digit(s, w_start, c);
      if (d) jac_madd(f, &buckets[d - 1], bx, bx + L);
    }
  }
  jac_t running;
  jac_set_inf(f, &running);
  for (size_t b = nb; b-- > 0;) {
    jac_add(f, &running, &buckets[b]);
    jac_add(f, &res, &running);
  }
  free(buckets);
  job->window_sums[wi] = res;
}

static void* msm_worker(void* arg) {
  msm_job_t* job = (msm_job_t*)arg;
  for (;;) {
    pthread_mutex_lock(&job->lock);
    int wi = job->next++;
    pthread_mutex_unlock(&job->lock);
    if (wi >= job->n_windows) break;
    msm_window(job, wi);
  }
  return NULL;
}

/* Returns 0 on success.  bases_xy: n * 2L u64 Montgomery affine; is_inf may be NULL; scalars canonical.
 * threads <= 1: serial (reference default features); else window-parallel (`parallel` feature). */
int ark_msm(int curve, const u64* bases_xy, const uint8_t* is_inf, const u64* scalars, size_t n, int threads,
            u64* out_xy, uint8_t* out_inf) {
  if (curve < 0 || curve > 1) return -1;
  curves_init();
  const curve_t* cv = &g_curves[curve];
  const field_t* f = &cv->fq;
  int L = f->n;
  u64* bases = (u64*)bases_xy;
  u64* owned = NULL;
  if (is_inf) {
    owned = (u64*)malloc((n ? n : 1) * 2 * L * 8);
    memcpy(owned, bases_xy, n * 2 * L * 8);
    for (size_t i = 0; i < n; i++)
      if (is_inf[i]) memset(owned + 2 * L * i, 0, 16 * L);
    bases = owned;
  }
  msm_job_t job;
  memset(&job, 0, sizeof(job));
  job.cv = cv;
  job.bases = bases;
  job.scalars = scalars;
  job.n = n;
  job.c = n < 32 ? 3 : ln_without_floats(n) + 2;
  job.n_windows = (cv->scalar_bits + job.c - 1) / job.c;
  job.window_sums = (jac_t*)malloc(job.n_windows * sizeof(jac_t));
  pthread_mutex_init(&job.lock, NULL);
  if (threads <= 1) {
    for (int w = 0; w < job.n_windows; w++) msm_window(&job, w);
  } else {
    if (threads > job.n_windows) threads = job.n_windows;
    pthread_t* th = (pthread_t*)malloc(threads * sizeof(pthread_t));
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, msm_worker, &job);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
  }
  /* lowest + fold from the highest: total = (total + sum_i) doubled c times */
  jac_t total;
  jac_set_inf(f, &total);
  for (int w = job.n_windows - 1; w >= 1; w--) {
    jac_add(f, &total, &job.window_sums[w]);
    for (int k = 0; k < job.c; k++) jac_dbl(f, &total);
  }
  jac_add(f, &total, &job.window_sums[0]);
  jac_to_affine(f, &total, out_xy, out_inf);
  free(job.window_sums);
  pthread_mutex_destroy(&job.lock);
  free(owned);
  return 0;
}

int ark_msm_window_bits(size_t n) { return n < 32 ? 3 : ln_without_floats(n) + 2; }

/* ---- synthetic inputs: same counter-based splitmix64 stream as oracle/pyref.py ------------------- */
static u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static u64 rng_word(u64 seed, u64 j) {
  return mix64(seed * 0xD1342543DE82EF95ull + j * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull);
}
void ark_rng_scalars(u64 seed, size_t n, u64* out) {
  for (size_t i = 0; i < n; i++) {
    for (int k = 0; k < 4; k++) out[4 * i + k] = rng_word(seed, 4 * i + k);
    out[4 * i + 3] &= ((u64)1 << 62) - 1;
  }
}

/* The SCALAR stream of amsm_vec_random since round 6 (accumulation_amd/csrc/rng.h:rng_scalar_fr): uniform in [0, r) of the
 * curve's scalar field by rejection -- candidate t of scalar i = words (t << 40) + 4 i .. + 3 masked to 255 bits, the first
 * one below r wins; after 64 rejections candidate 63 with bit 254 cleared.  ark_rng_scalars above stays the 254-bit
 * MULTIPLIER stream of the synthetic committer keys (G_i = k_i G). */
int ark_rng_scalars_fr(int curve, u64 seed, size_t n, u64* out) {
  if (curve < 0 || curve > 1) return -1;
  curves_init();
  const field_t* fr = &g_curves[curve].fr;
  for (size_t i = 0; i < n; i++) {
    u64* o = out + 4 * i;
    int ok = 0;
    for (u64 t = 0; t < 64 && !ok; t++) {
      for (int k = 0; k < 4; k++) o[k] = rng_word(seed, (t << 40) + 4 * i + k);
      o[3] &= ((u64)1 << 63) - 1;
      ok = !geq(o, fr->m, 4);
    }
    if (!ok) o[3] &= ((u64)1 << 62) - 1;
  }
  return 0;
}

typedef struct {
  const curve_t* cv;
  u64 seed;
  size_t lo, hi;
  u64* out;
  const u64* table; /* 8-bit fixed-base table, see ark_rng_points */
} gen_job_t;

/* fixed-base table: T[w][d] = d * 2^(8w) * G, d = 1..255, affine Montgomery */
static void* gen_worker(void* arg) {
  gen_job_t* j = (gen_job_t*)arg;
  const field_t* f = &j->cv->fq;
  int L = f->n;
  for (size_t i = j->lo; i < j->hi; i++) {
    u64 k[4];
    for (int q = 0; q < 4; q++) k[q] = rng_word(j->seed, 4 * i + q);
    k[3] &= ((u64)1 << 62) - 1;
    jac_t acc;
    jac_set_inf(f, &acc);
    for (int w = 0; w < 32; w++) {
      unsigned d = (unsigned)((k[w >> 3] >> ((w & 7) * 8)) & 0xff);
      if (d) {
        const u64* e = j->table + ((size_t)w * 255 + (d - 1)) * 2 * L;
        jac_madd(f, &acc, e, e + L);
      }
    }
    uint8_t inf;
    jac_to_affine(f, &acc, j->out + 2 * L * i, &inf);
  }
  return NULL;
}

/* out: n * 2L u64; P_i = rng_scalar(seed, i) * G */
int ark_rng_points(int curve, u64 seed, size_t n, int threads, u64* out) {
  if (curve < 0 || curve > 1) return -1;
  curves_init();
  const curve_t* cv = &g_curves[curve];
  const field_t* f = &cv->fq;
  int L = f->n;
  u64* table = (u64*)malloc((size_t)32 * 255 * 2 * L * 8);
  u64 gx[MAXL], gy[MAXL];
  f_to_mont(f, gx, cv->gx);
  f_to_mont(f, gy, cv->gy);
  jac_t base;
  memcpy(base.x, gx, 8 * L);
  memcpy(base.y, gy, 8 * L);
  memcpy(base.z, f->one, 8 * L);
  for (int w = 0; w < 32; w++) {
    jac_t acc;
    jac_set_inf(f, &acc);
    for (int d = 1; d <= 255; d++) {
      jac_add(f, &acc, &base);
      uint8_t inf;
      jac_to_affine(f, &acc, table + ((size_t)w * 255 + (d - 1)) * 2 * L, &inf);
    }
    for (int k = 0; k < 8; k++) jac_dbl(f, &base);
  }
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  pthread_t* th = (pthread_t*)malloc(threads * sizeof(pthread_t));
  gen_job_t* jobs = (gen_job_t*)malloc(threads * sizeof(gen_job_t));
  for (int t = 0; t < threads; t++) {
    jobs[t].cv = cv;
    jobs[t].seed = seed;
    jobs[t].lo = n * t / threads;
    jobs[t].hi = n * (t + 1) / threads;
    jobs[t].out = out;
    jobs[t].table = table;
    pthread_create(&th[t], NULL, gen_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
  free(table);
  return 0;
}

/* ---- the IPA opening's folds (ark_poly_commit::ipa_pc, ext: `key_l += key_r.mul(round_challenge)` per round, under
 * src/ipa_pc_as/mod.rs:454; restated from the published construction, BCMS20 section 7 / the Halo-style IPA) -------------
 * out[i] = l[i] + x * r[i] over affine Montgomery points ((0, 0) = identity), x canonical (4 words): plain left-to-right
 * double-and-add in Jacobian coordinates, one inversion per point -- the definition, nothing clever, so that the device's
 * three fold formulations (NAF ladder, joint ladder over window multiples, GLV split) have an independent check at the
 * opening's sizes. */
typedef struct {
  const curve_t* cv;
  const u64 *l, *r, *x;
  size_t lo, hi;
  u64* out;
} fold_job_t;
static void* fold_worker(void* arg) {
  fold_job_t* j = (fold_job_t*)arg;
  const field_t* f = &j->cv->fq;
  int L = f->n;
  int top = -1;
  for (int b = 255; b >= 0; b--)
    if ((j->x[b >> 6] >> (b & 63)) & 1) {
      top = b;
      break;
    }
  for (size_t i = j->lo; i < j->hi; i++) {
    const u64 *lx = j->l + 2 * L * i, *rx = j->r + 2 * L * i;
    jac_t acc;
    jac_set_inf(f, &acc);
    int r_inf = is_zero(rx, 2 * L);
    if (!r_inf)
      for (int b = top; b >= 0; b--) {
        jac_dbl(f, &acc);
        if ((j->x[b >> 6] >> (b & 63)) & 1) jac_madd(f, &acc, rx, rx + L);
      }
    if (!is_zero(lx, 2 * L)) jac_madd(f, &acc, lx, lx + L);
    uint8_t inf;
    jac_to_affine(f, &acc, j->out + 2 * L * i, &inf);
  }
  return NULL;
}
int ark_points_fold(int curve, const u64* l_xy, const u64* r_xy, size_t n, const u64* x_canon, int threads, u64* out_xy) {
  if (curve < 0 || curve > 1) return -1;
  curves_init();
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  pthread_t* th = (pthread_t*)malloc(threads * sizeof(pthread_t));
  fold_job_t* jobs = (fold_job_t*)malloc(threads * sizeof(fold_job_t));
  for (int t = 0; t < threads; t++) {
    jobs[t].cv = &g_curves[curve];
    jobs[t].l = l_xy;
    jobs[t].r = r_xy;
    jobs[t].x = x_canon;
    jobs[t].lo = n * t / threads;
    jobs[t].hi = n * (t + 1) / threads;
    jobs[t].out = out_xy;
    pthread_create(&th[t], NULL, fold_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
  return 0;
}

/* ---- scalar-field vector loops (Montgomery in, Montgomery out) --------------------------------- */
/* compute_hp, src/hp_as/mod.rs:278-285 */
int ark_fr_hadamard(int curve, const u64* a, const u64* b, size_t n, u64* out) {
  if (curve < 0 || curve > 1) return -1;
  curves_init();
  const field_t* f = &g_curves[curve].fr;
  for (size_t i = 0; i < n; i++) f_mul(f, out + 4 * i, a + 4 * i, b + 4 * i);
  return 0;
}
/* out[i] = sum_j coeff[j] * vecs[j][i] (+ hiding[i]); combine_vectors, src/hp_as/mod.rs:492-512 */
int ark_fr_combine(int curve, const u64* const* vecs, const size_t* lens, size_t n_vecs, const u64* coeffs,
                   const u64* hiding, size_t hiding_len, size_t n, u64* out) {
  if (curve < 0 || curve > 1) return -1;
  curves_init();
  const field_t* f = &g_curves[curve].fr;
  for (size_t i = 0; i < n; i++) {
    u64 acc[4] = {0, 0, 0, 0}, t[4];
    if (hiding && i < hiding_len) memcpy(acc, hiding + 4 * i, 32);
    for (size_t j = 0; j < n_vecs; j++) {
      if (i >= lens[j]) continue;
      f_mul(f, t, coeffs + 4 * j, vecs[j] + 4 * i);
      f_add(f, acc, acc, t);
    }
    memcpy(out + 4 * i, acc, 32);
  }
  return 0;
}
/* <a, b> (Montgomery in, Montgomery out): the opening's inner products and evaluations */
int ark_fr_inner_product(int curve, const u64* a, const u64* b, size_t n, u64* out) {
  if (curve < 0 || curve > 1) return -1;
  curves_init();
  const field_t* f = &g_curves[curve].fr;
  u64 acc[4] = {0, 0, 0, 0}, t[4];
  for (size_t i = 0; i < n; i++) {
    f_mul(f, t, a + 4 * i, b + 4 * i);
    f_add(f, acc, acc, t);
  }
  memcpy(out, acc, 32);
  return 0;
}
/* out[i] = point^i (Montgomery in, Montgomery out): the evaluation vector z of the opening */
int ark_fr_powers(int curve, const u64* point, size_t n, u64* out) {
  if (curve < 0 || curve > 1) return -1;
  curves_init();
  const field_t* f = &g_curves[curve].fr;
  u64 cur[4];
  memcpy(cur, f->one, 32);
  for (size_t i = 0; i < n; i++) {
    memcpy(out + 4 * i, cur, 32);
    f_mul(f, cur, cur, point);
  }
  return 0;
}
int ark_fr_to_mont(int curve, const u64* a, size_t n, u64* out) {
  if (curve < 0 || curve > 1) return -1;
  curves_init();
  const field_t* f = &g_curves[curve].fr;
  for (size_t i = 0; i < n; i++) f_to_mont(f, out + 4 * i, a + 4 * i);
  return 0;
}
int ark_fr_from_mont(int curve, const u64* a, size_t n, u64* out) {
  if (curve < 0 || curve > 1) return -1;
  curves_init();
  const field_t* f = &g_curves[curve].fr;
  for (size_t i = 0; i < n; i++) f_from_mont(f, out + 4 * i, a + 4 * i);
  return 0;
}

/* compute_t_vecs, src/hp_as/mod.rs:288-349 (Montgomery in, Montgomery out): per index li the coefficients of
 * (sum_j mu[j] a_j[li] X^j [+ hiding_a[li] mu[n]]) * (sum_j b_{n-1-j}[li] X^j [+ hiding_b[li] mu[1]]), missing entries
 * read as zero (:306-318), the hiding terms added to coefficient 0 of each factor (:322-330), naive_mul (:335).
 * a/b: n pointers + lengths; mu: n (+1 with hiding) elements; out: 2n-1 pointers to hp_len elements each. */
#define ARK_T_MAX 16
int ark_fr_t_vecs(int curve, const u64* const* a, const size_t* a_lens, const u64* const* b, const size_t* b_lens,
                  size_t n, const u64* mu, size_t hp_len, const u64* hiding_a, size_t ha_len, const u64* hiding_b,
                  size_t hb_len, u64* const* out) {
  if (curve < 0 || curve > 1 || n == 0 || n > ARK_T_MAX) return -1;
  curves_init();
  const field_t* f = &g_curves[curve].fr;
  const int hiding = hiding_a != NULL && hiding_b != NULL;
  for (size_t li = 0; li < hp_len; li++) {
    u64 ac[ARK_T_MAX][4], bc[ARK_T_MAX][4], t[2 * ARK_T_MAX][4], tmp[4];
    for (size_t j = 0; j < n; j++) {
      if (li < a_lens[j]) f_mul(f, ac[j], mu + 4 * j, a[j] + 4 * li);
      else memset(ac[j], 0, 32);
      /* b_coeffs.reverse() (:320): position j holds witness n-1-j */
      size_t src = n - 1 - j;
      if (li < b_lens[src]) memcpy(bc[j], b[src] + 4 * li, 32);
      else memset(bc[j], 0, 32);
    }
    if (hiding) {
      if (li < ha_len) {
        f_mul(f, tmp, hiding_a + 4 * li, mu + 4 * n);
        f_add(f, ac[0], ac[0], tmp);
      }
      if (li < hb_len) {
        f_mul(f, tmp, hiding_b + 4 * li, mu + 4 * 1);
        f_add(f, bc[0], bc[0], tmp);
      }
    }
    memset(t, 0, sizeof(t));
    for (size_t i = 0; i < n; i++)
      for (size_t j = 0; j < n; j++) {
        f_mul(f, tmp, ac[i], bc[j]);
        f_add(f, t[i + j], t[i + j], tmp);
      }
    for (size_t k = 0; k < 2 * n - 1; k++) memcpy(out[k] + 4 * li, t[k], 32);
  }
  return 0;
}

/* matrix_vec_mul / inner_prod, src/r1cs_nark_as/r1cs_nark/mod.rs:443-462: out[r] = sum_k coeff[k] * z[col[k]] over the
 * entries row_ptr[r] .. row_ptr[r+1] of a row-sparse matrix; z = input || witness, given as two arrays (index i reads
 * input[i] if i < n_input else witness[i - n_input], :452-456).  Montgomery in, Montgomery out. */
int ark_fr_spmv(int curve, const u64* row_ptr, const u64* col, const u64* coeff, size_t n_rows, const u64* input,
                size_t n_input, const u64* witness, size_t n_witness, u64* out) {
  if (curve < 0 || curve > 1) return -1;
  curves_init();
  const field_t* f = &g_curves[curve].fr;
  for (size_t r = 0; r < n_rows; r++) {
    u64 acc[4] = {0, 0, 0, 0}, t[4];
    for (u64 k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
      u64 i = col[k];
      if (i >= n_input + n_witness) return -2;
      const u64* z = i < n_input ? input + 4 * i : witness + 4 * (i - n_input);
      f_mul(f, t, z, coeff + 4 * k);
      f_add(f, acc, acc, t);
    }
    memcpy(out + 4 * r, acc, 32);
  }
  return 0;
}

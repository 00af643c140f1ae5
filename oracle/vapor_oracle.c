/*
 * vapor_oracle.c - CPU restatement of VaPoR's recurrence-plot scoring path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vapor_amd/ may link, load or call this
 * file; it exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg have an independent checker for the HIP path.
 *
 * Parity status: PINNED - every function below is checked against vectors that the
 * reference itself produced in the development container (oracle/gen_golden.py ->
 * tests/golden/ *.json.gz, tests/test_oracle_golden.py).
 *
 * Citations are file:line in /root/reference, SF = vapor_vali/Simple_function.pyx.
 *
 * The reference is untyped Python: dict-of-lists hash join (SF:951-983) and list
 * membership scans in the cleaners (SF:551-580).  This file keeps the results and
 * orders of those routines and replaces only the O(n*m) membership scans by a sort +
 * binary search over the same value groups.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VO_OK 0
#define VO_OVERFLOW (-2)
#define VO_KEYERROR (-3)
#define VO_NOMEM (-5)

/* key_modify, SF:908-949: IUPAC ambiguity codes fold to N (case kept). */
static unsigned char vo_fold(unsigned char c)
{
    switch (c) {
    case 'R': case 'Y': case 'S': case 'W': case 'K': case 'M': case 'B': case 'D': case 'H': case 'V':
        return 'N';
    case 'r': case 'y': case 's': case 'w': case 'k': case 'm': case 'b': case 'd': case 'h': case 'v':
        return 'n';
    default:
        return c;
    }
}

/* invert_base, SF:19-20; -1 = KeyError. */
static int vo_comp(unsigned char c)
{
    switch (c) {
    case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; case 'N': return 'N';
    case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c'; case 'n': return 'n';
    default: return -1;
    }
}

static uint64_t vo_hash(const unsigned char *p, int k)
{
    uint64_t h = 1469598103934665603ULL;
    for (int t = 0; t < k; ++t) {
        h ^= p[t];
        h *= 1099511628211ULL;
    }
    return h ^ (h >> 29);
}

/*
 * dotdata(kmerlen, seq1, seq2) = kmerhits(seq1, seq2, kmerlen, 1, True), SF:545-549, 951-983.
 * Emits (j, i) with j = position in seq2 ascending, and for one j the matching seq1
 * positions in the order the reference appended them to lookup[key]: i ascending,
 * forward entry before reverse-complement entry (so a k-mer equal to its own reverse
 * complement yields the tuple twice).  k > 40 (edit-distance branch, SF:969-973) is not
 * reachable from window_size_refine and is not restated.
 * Returns VO_KEYERROR where the reference raises KeyError (SF:1421): some k-mer of
 * seq1 holds a character invert_base lacks after folding.
 * hits_ji may be NULL (count only).  On VO_OVERFLOW *n_hits is the required count.
 */
int vo_dotdata(int k, const char *s1, int n1, const char *s2, int n2,
               int32_t *hits_ji, int64_t cap, int64_t *n_hits)
{
    *n_hits = 0;
    int nk1 = n1 - k + 1, nk2 = n2 - k + 1;
    if (nk1 <= 0)
        return VO_OK;              /* empty lookup, no subkeys() call, no KeyError */
    unsigned char *f1 = (unsigned char *)malloc((size_t)n1 * 2 + 2);
    if (!f1) return VO_NOMEM;
    unsigned char *rc1 = f1 + n1 + 1;
    for (int p = 0; p < n1; ++p) {
        unsigned char c = vo_fold((unsigned char)s1[p]);
        int cc = vo_comp(c);
        if (cc < 0) { free(f1); return VO_KEYERROR; }
        f1[p] = c;
        rc1[n1 - 1 - p] = (unsigned char)cc;
    }
    if (nk2 <= 0) { free(f1); return VO_OK; }
    int64_t ne = 2 * (int64_t)nk1;
    uint64_t H = 1;
    while ((int64_t)H < 2 * ne) H <<= 1;
    int32_t *head = (int32_t *)malloc(sizeof(int32_t) * (H + (size_t)ne));
    unsigned char *f2 = (unsigned char *)malloc((size_t)n2 + 1);
    if (!head || !f2) { free(f1); free(head); free(f2); return VO_NOMEM; }
    int32_t *next = head + H;
    memset(head, 0xff, sizeof(int32_t) * H);
    for (int64_t e = ne - 1; e >= 0; --e) {
        int i = (int)(e >> 1);
        const unsigned char *key = (e & 1) ? rc1 + (n1 - k - i) : f1 + i;
        uint64_t h = vo_hash(key, k) & (H - 1);
        next[e] = head[h];
        head[h] = (int32_t)e;      /* pushing from the back leaves each chain ascending in e */
    }
    for (int p = 0; p < n2; ++p) f2[p] = vo_fold((unsigned char)s2[p]);
    int64_t n = 0;
    for (int j = 0; j < nk2; ++j) {
        const unsigned char *q = f2 + j;
        uint64_t h = vo_hash(q, k) & (H - 1);
        for (int32_t e = head[h]; e >= 0; e = next[e]) {
            int i = e >> 1;
            const unsigned char *key = (e & 1) ? rc1 + (n1 - k - i) : f1 + i;
            if (memcmp(key, q, (size_t)k) == 0) {
                if (hits_ji && n < cap) { hits_ji[2 * n] = j; hits_ji[2 * n + 1] = i; }
                ++n;
            }
        }
    }
    free(f1); free(head); free(f2);
    *n_hits = n;
    if (hits_ji && n > cap) return VO_OVERFLOW;
    return VO_OK;
}

/* ------------------------------------------------------------------------------------
 * 1-D gap clustering shared by dis_cluster (SF:551-564) and dis_cluster_2 (SF:566-580):
 * sort the values, start a new group whenever a value exceeds its predecessor by >= 10.
 * size_of[t] receives the size of the group holding v[t]; *max_size the largest group.
 */
static int vo_cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

static int vo_group_sizes(const int32_t *v, int64_t n, int32_t *size_of, int32_t *max_size)
{
    *max_size = 0;
    if (n <= 0) return VO_OK;
    int32_t *s = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 3);
    if (!s) return VO_NOMEM;
    int32_t *gstart = s + n, *gsize = s + 2 * n;
    memcpy(s, v, sizeof(int32_t) * (size_t)n);
    qsort(s, (size_t)n, sizeof(int32_t), vo_cmp_i32);
    int64_t ng = 0;
    gstart[0] = s[0]; gsize[0] = 1;
    for (int64_t t = 1; t < n; ++t) {
        if (s[t] - s[t - 1] < 10) gsize[ng]++;
        else { ++ng; gstart[ng] = s[t]; gsize[ng] = 1; }
    }
    ++ng;
    for (int64_t g = 0; g < ng; ++g)
        if (gsize[g] > *max_size) *max_size = gsize[g];
    for (int64_t t = 0; t < n; ++t) {
        int64_t lo = 0, hi = ng - 1;           /* last group whose first value <= v[t] */
        while (lo < hi) {
            int64_t mid = (lo + hi + 1) >> 1;
            if (gstart[mid] <= v[t]) lo = mid; else hi = mid - 1;
        }
        size_of[t] = gsize[lo];
    }
    free(s);
    return VO_OK;
}

/*
 * clean_dotdata_diagnal_and_anti_diagnal, SF:432-448 with dis_cluster_2 (groups of more
 * than 10 survive): a dot is dropped only when its i-j group AND its i+j group are
 * both dropped.  keep[t] = 1/0 in input order.
 */
int vo_clean_c1(const int32_t *hits_ji, int64_t n, uint8_t *keep)
{
    if (n <= 0) return VO_OK;
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 4);
    if (!buf) return VO_NOMEM;
    int32_t *d = buf, *a = buf + n, *sd = buf + 2 * n, *sa = buf + 3 * n, mx;
    for (int64_t t = 0; t < n; ++t) {
        d[t] = hits_ji[2 * t + 1] - hits_ji[2 * t];
        a[t] = hits_ji[2 * t + 1] + hits_ji[2 * t];
    }
    int rc = vo_group_sizes(d, n, sd, &mx);
    if (rc == VO_OK) rc = vo_group_sizes(a, n, sa, &mx);
    if (rc == VO_OK)
        for (int64_t t = 0; t < n; ++t)
            keep[t] = !(sd[t] <= 10 && sa[t] <= 10);
    free(buf);
    return rc;
}

/*
 * The cleaning inside calcu_vapor_single_read_score_within_10Perc_m1b, SF:281-288:
 * clean_dotdata_diagnal_m1b (SF:404-416) on i-j over all dots, then
 * clean_dotdata_anti_diagnal_m1b (SF:418-430) on i+j over the dots the first step did
 * not keep, both with dis_cluster's rule (SF:560-563): groups of more than 50 survive;
 * if there is none, every group of maximal size survives.
 * keep[t] = 1 (kept by the diagonal step), 2 (kept by the anti-diagonal step), 0.
 */
int vo_clean_c2(const int32_t *hits_ji, int64_t n, uint8_t *keep)
{
    if (n <= 0) return VO_OK;
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 3);
    if (!buf) return VO_NOMEM;
    int32_t *v = buf, *sz = buf + n, *idx = buf + 2 * n, mx;
    for (int64_t t = 0; t < n; ++t) v[t] = hits_ji[2 * t + 1] - hits_ji[2 * t];
    int rc = vo_group_sizes(v, n, sz, &mx);
    if (rc != VO_OK) { free(buf); return rc; }
    int64_t m = 0;
    for (int64_t t = 0; t < n; ++t) {
        keep[t] = (mx > 50) ? (sz[t] > 50) : (sz[t] == mx);
        if (!keep[t]) idx[m++] = (int32_t)t;
    }
    if (m > 0) {
        for (int64_t u = 0; u < m; ++u) v[u] = hits_ji[2 * idx[u] + 1] + hits_ji[2 * idx[u]];
        rc = vo_group_sizes(v, m, sz, &mx);
        if (rc == VO_OK)
            for (int64_t u = 0; u < m; ++u)
                if ((mx > 50) ? (sz[u] > 50) : (sz[u] == mx)) keep[idx[u]] = 2;
    }
    free(buf);
    return rc;
}

/*
 * Per-pair integer sufficient statistics, the same record the HIP library returns
 * (include/vapor_hip.h, VAPOR_ST_*), computed the reference's way:
 *  [0] n_hits            len(dotdata)
 *  [1] first_j           dotdata[0][0]   (-1 when empty)          SF:187
 *  [2] last_j            dotdata[-1][0]  (-1 when empty)
 *  [3] c1_kept           len(clean_dotdata_diagnal_and_anti_diagnal)            SF:432-448
 *  [4] c1_sum_abs        sum(abs(j-i)) over the C1-kept dots (eu_dis_abs_calcu) SF:705-708
 *  [5] c2_kept           dots kept by the two-step C2 cleaning                  SF:281-288
 *  [6] c2_count10        eu_dis_dots_within_10perc over them: j>0 and abs(j-i)/j < 0.16,
 *                        i.e. 25*abs(j-i) < 4*j                                 SF:730-733
 *  [7] n_diag            dots with j == i            (qual_check_repetitive_region SF:1158-1160)
 *  [8] n_lower           dots with j >  i                                        SF:1162-1164
 *  [9] c2_kept_diag      dots kept by the diagonal step alone
 * keep1 / keep2 (n_hits bytes each, may be NULL) receive the C1 / C2 flags.
 */
int vo_pair_stats(int k, const char *s1, int n1, const char *s2, int n2,
                  int64_t *st, int32_t *hits_ji, int64_t cap, uint8_t *keep1, uint8_t *keep2)
{
    int64_t n = 0;
    for (int t = 0; t < 16; ++t) st[t] = 0;
    st[1] = st[2] = -1;
    int rc = vo_dotdata(k, s1, n1, s2, n2, hits_ji, cap, &n);
    st[0] = n;
    if (rc != VO_OK) return rc;
    if (n == 0) return VO_OK;
    uint8_t *k1 = keep1 ? keep1 : (uint8_t *)malloc((size_t)n);
    uint8_t *k2 = keep2 ? keep2 : (uint8_t *)malloc((size_t)n);
    if (!k1 || !k2) return VO_NOMEM;
    rc = vo_clean_c1(hits_ji, n, k1);
    if (rc == VO_OK) rc = vo_clean_c2(hits_ji, n, k2);
    if (rc == VO_OK) {
        st[1] = hits_ji[0];
        st[2] = hits_ji[2 * (n - 1)];
        for (int64_t t = 0; t < n; ++t) {
            int64_t j = hits_ji[2 * t], i = hits_ji[2 * t + 1];
            int64_t ad = j > i ? j - i : i - j;
            if (k1[t]) { st[3]++; st[4] += ad; }
            if (k2[t]) { st[5]++; if (j > 0 && 25 * ad < 4 * j) st[6]++; }
            if (k2[t] == 1) st[9]++;
            if (j == i) st[7]++;
            else if (j > i) st[8]++;
        }
    }
    if (!keep1) free(k1);
    if (!keep2) free(k2);
    return rc;
}

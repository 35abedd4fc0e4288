// libsvm_reader.cpp — native replacement of the reference's Python line parser (data_loader.py:12-47):
// `label id:val id:val ...` text  ->  int64 ids [N,F], float vals [N,F], float y [N].   Host code (g++).
//
// Semantics kept from LibsvmDataset.__init__:
//   * a line is split on single spaces; column 0 is the label (float), the others are `id:value`
//     (int(id), float(value));
//   * a line that fails to parse, or does not have exactly nfields pairs, is SKIPPED (the reference's
//     try/except prints it and continues) and the following samples move up: output rows are the valid
//     lines in file order;  nsamples = number of valid lines.
// The file is mmap-ed and parsed by OpenMP threads over line ranges; numbers go through strtoll/strtof on a
// bounded copy of the token (Python accepts a few more spellings — underscores, surrounding blanks — which
// real libsvm files do not contain; such lines would be skipped here).
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif

namespace {

struct Mapped {
    const char* p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = ""; return true; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const char*)m;
        madvise(m, n, MADV_SEQUENTIAL);
        return true;
    }
    ~Mapped() {
        if (p && n) munmap((void*)p, n);
        if (fd >= 0) close(fd);
    }
};

inline bool parse_int(const char* b, const char* e, int64_t& out) {
    // Python int(): optional sign, digits only
    if (b == e) return false;
    const char* q = b;
    if (*q == '+' || *q == '-') ++q;
    if (q == e) return false;
    for (const char* t = q; t < e; ++t)
        if (*t < '0' || *t > '9') return false;
    char buf[32];
    const size_t len = (size_t)(e - b);
    if (len >= sizeof(buf)) return false;
    memcpy(buf, b, len);
    buf[len] = 0;
    errno = 0;
    out = strtoll(buf, nullptr, 10);
    return errno == 0;
}

inline bool parse_float(const char* b, const char* e, float& out) {
    // Python float(): strips surrounding whitespace (the last column carries the newline)
    while (b < e && (*b == ' ' || *b == '\t' || *b == '\r' || *b == '\n')) ++b;
    while (e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r' || e[-1] == '\n')) --e;
    if (b == e) return false;
    char buf[64];
    const size_t len = (size_t)(e - b);
    if (len >= sizeof(buf)) return false;
    memcpy(buf, b, len);
    buf[len] = 0;
    if ((buf[0] == '0' && (buf[1] == 'x' || buf[1] == 'X'))) return false;   // strtod hex floats: not Python
    char* end = nullptr;
    const double d = strtod(buf, &end);      // Python parses to double, FloatTensor rounds to fp32
    if (end != buf + len) return false;
    out = (float)d;
    return true;
}

// one line [b, e) WITHOUT its trailing '\n'
inline bool parse_line(const char* b, const char* e, int nfields, int64_t* ids, float* vals, float* y) {
    const char* sp = (const char*)memchr(b, ' ', (size_t)(e - b));
    const char* lab_end = sp ? sp : e;
    if (!parse_float(b, lab_end, *y)) return false;
    int k = 0;
    const char* q = sp ? sp + 1 : e;
    if (!sp) return nfields == 0;
    while (true) {
        const char* nx = (const char*)memchr(q, ' ', (size_t)(e - q));
        const char* ce = nx ? nx : e;
        const char* colon = (const char*)memchr(q, ':', (size_t)(ce - q));
        if (!colon) return false;
        // Python: col.split(':') then pair[0], pair[1]; extra ':' parts are ignored by the map_func
        const char* colon2 = (const char*)memchr(colon + 1, ':', (size_t)(ce - colon - 1));
        const char* ve = colon2 ? colon2 : ce;
        if (k >= nfields) return false;
        if (!parse_int(q, colon, ids[k])) return false;
        if (!parse_float(colon + 1, ve, vals[k])) return false;
        ++k;
        if (!nx) break;
        q = nx + 1;
    }
    return k == nfields;
}

}  // namespace

extern "C" {

// number of lines as `sum(1 for line in f)` counts them (data_loader.py:25-26)
int64_t armnet_libsvm_count_lines(const char* path) {
    Mapped m;
    if (!m.open(path)) return -1;
    int64_t n = 0;
    for (size_t i = 0; i < m.n; ++i) n += m.p[i] == '\n';
    if (m.n && m.p[m.n - 1] != '\n') ++n;
    return n;
}

// ids [cap, nfields], vals [cap, nfields], y [cap] with cap >= count_lines(path).
// Returns nsamples (valid lines, compacted in file order) or -1 on I/O error; *n_bad = skipped lines.
int64_t armnet_libsvm_parse(const char* path, int nfields, int64_t cap, int64_t* ids, float* vals, float* y,
                            int64_t* n_bad, int nthreads) {
    Mapped m;
    if (!m.open(path)) return -1;
    // line starts
    std::vector<size_t> starts;
    starts.reserve(m.n / 32 + 16);
    size_t pos = 0;
    while (pos < m.n) {
        starts.push_back(pos);
        const char* nl = (const char*)memchr(m.p + pos, '\n', m.n - pos);
        pos = nl ? (size_t)(nl - m.p) + 1 : m.n;
    }
    const int64_t nlines = (int64_t)starts.size();
    if (nlines > cap) return -2;
    std::vector<uint8_t> ok((size_t)nlines, 0);
#if defined(_OPENMP)
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nlines; ++i) {
        const char* b = m.p + starts[(size_t)i];
        const char* e = (i + 1 < nlines) ? m.p + starts[(size_t)i + 1] : m.p + m.n;
        if (e > b && e[-1] == '\n') --e;
        ok[(size_t)i] = parse_line(b, e, nfields, ids + i * nfields, vals + i * nfields, y + i) ? 1 : 0;
    }
    // compact the valid rows in file order
    int64_t w = 0;
    for (int64_t i = 0; i < nlines; ++i) {
        if (!ok[(size_t)i]) continue;
        if (w != i) {
            memmove(ids + w * nfields, ids + i * nfields, sizeof(int64_t) * (size_t)nfields);
            memmove(vals + w * nfields, vals + i * nfields, sizeof(float) * (size_t)nfields);
            y[w] = y[i];
        }
        ++w;
    }
    if (n_bad) *n_bad = nlines - w;
    return w;
}

}  // extern "C"

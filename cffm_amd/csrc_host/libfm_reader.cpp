// Fast libfm reader (host C++): the N2 "next" row of SURVEY section 8(f).
//
// Reproduces /root/reference LoadData.py:33-103 exactly: the dictionary key is the WHOLE token "idx:val" (the value
// part is never parsed), ids are handed out in first-appearance order over the files in the order given (the Python
// side passes train, test, validation - LoadData.py:35-39), features_M = number of distinct tokens, labels are
// float(items[0]).  One pass per file over an mmap of the text, an open-addressing hash table keyed by the token
// bytes, output as packed arrays (ids row-major with a row-offset table, so ragged rows survive).
//
// C ABI (bound with ctypes from cffm_amd/LoadData.py):
//   h = libfm_open();  libfm_read_file(h, path) for each file;  query sizes;  copy out;  libfm_close(h).
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

namespace {

struct Split {
    std::vector<double> y;
    std::vector<int32_t> ids;
    std::vector<int64_t> row_off;   // size rows + 1
};

struct Reader {
    // open addressing: slot -> token index (+1), 0 = empty
    std::vector<uint32_t> table;
    std::vector<uint64_t> hashes;          // per token
    std::vector<uint32_t> tok_off;         // offset of token i in arena, size n+1
    std::string arena;                     // token bytes, concatenated
    std::vector<Split> splits;
    std::vector<int64_t> m_after;          // dictionary size after each file (the reference prints it)

    Reader() : table(1u << 16, 0u) { tok_off.push_back(0); }

    static uint64_t hash(const char* p, size_t n) {      // FNV-1a
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
        return h;
    }
    void grow() {
        std::vector<uint32_t> nt(table.size() * 2, 0u);
        const size_t mask = nt.size() - 1;
        for (uint32_t i = 0; i < hashes.size(); ++i) {
            size_t s = hashes[i] & mask;
            while (nt[s]) s = (s + 1) & mask;
            nt[s] = i + 1;
        }
        table.swap(nt);
    }
    int32_t lookup_or_add(const char* p, size_t n) {
        const uint64_t h = hash(p, n);
        size_t mask = table.size() - 1, s = h & mask;
        while (table[s]) {
            const uint32_t t = table[s] - 1;
            if (hashes[t] == h && tok_off[t + 1] - tok_off[t] == n && memcmp(arena.data() + tok_off[t], p, n) == 0) return (int32_t)t;
            s = (s + 1) & mask;
        }
        const uint32_t id = (uint32_t)hashes.size();
        hashes.push_back(h);
        arena.append(p, n);
        tok_off.push_back((uint32_t)arena.size());
        table[s] = id + 1;
        if (hashes.size() * 2 > table.size()) grow();
        return (int32_t)id;
    }
};

}  // namespace

extern "C" {

void* libfm_open(void) { return new Reader(); }
void libfm_close(void* h) { delete (Reader*)h; }

// returns the split index (>= 0) or -1 on I/O error
int libfm_read_file(void* h, const char* path) {
    Reader* r = (Reader*)h;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return -1; }
    const size_t len = (size_t)st.st_size;
    const char* base = len ? (const char*)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    if (len && base == MAP_FAILED) { close(fd); return -1; }
    Split sp;
    sp.row_off.push_back(0);
    const char *p = base, *end = base + len;
    while (p < end) {
        const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        // Python: line.strip().split(' ') - strip both ends of whitespace, split on single spaces
        const char *a = p, *b = eol;
        while (a < b && (*a == ' ' || *a == '\t' || *a == '\r' || *a == '\f' || *a == '\v')) ++a;
        while (b > a && (b[-1] == ' ' || b[-1] == '\t' || b[-1] == '\r' || b[-1] == '\f' || b[-1] == '\v')) --b;
        // (a blank line in the middle of a file would make the reference raise at float(''); files end at EOF here)
        if (a < b || eol < end) {
            const char* q = (const char*)memchr(a, ' ', (size_t)(b - a));
            const char* lab_end = q ? q : b;
            sp.y.push_back(strtod(std::string(a, lab_end).c_str(), nullptr));
            const char* t = q ? q + 1 : b;
            while (t <= b && q) {
                const char* nx = (const char*)memchr(t, ' ', (size_t)(b - t));
                const char* te = nx ? nx : b;
                sp.ids.push_back(r->lookup_or_add(t, (size_t)(te - t)));      // empty tokens (double spaces) are keys too, as in Python
                if (!nx) break;
                t = nx + 1;
            }
            sp.row_off.push_back((int64_t)sp.ids.size());
        }
        p = eol + 1;
    }
    if (len) munmap((void*)base, len);
    close(fd);
    r->splits.push_back(std::move(sp));
    r->m_after.push_back((int64_t)r->hashes.size());
    return (int)r->splits.size() - 1;
}

int64_t libfm_num_features(void* h) { return (int64_t)((Reader*)h)->hashes.size(); }
int64_t libfm_features_after(void* h, int split) { return ((Reader*)h)->m_after[split]; }
int64_t libfm_rows(void* h, int split) { return (int64_t)((Reader*)h)->splits[split].y.size(); }
int64_t libfm_nnz(void* h, int split) { return (int64_t)((Reader*)h)->splits[split].ids.size(); }
void libfm_copy_split(void* h, int split, double* y, int32_t* ids, int64_t* row_off) {
    const Split& s = ((Reader*)h)->splits[split];
    memcpy(y, s.y.data(), s.y.size() * sizeof(double));
    memcpy(ids, s.ids.data(), s.ids.size() * sizeof(int32_t));
    memcpy(row_off, s.row_off.data(), s.row_off.size() * sizeof(int64_t));
}
int64_t libfm_arena_bytes(void* h) { return (int64_t)((Reader*)h)->arena.size(); }
void libfm_copy_tokens(void* h, char* arena, uint32_t* tok_off) {
    Reader* r = (Reader*)h;
    memcpy(arena, r->arena.data(), r->arena.size());
    memcpy(tok_off, r->tok_off.data(), r->tok_off.size() * sizeof(uint32_t));
}

}  // extern "C"

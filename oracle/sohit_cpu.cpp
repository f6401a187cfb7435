// sohit_cpu -- CPU restatement of SwiftOrtho's seed-and-extend hot path.
//
// TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path in
// swiftortho_amd/csrc and the timed "cpu_baseline" of bench.py.  Nothing in the
// product (swiftortho_amd/, bin/, libsohit.so) may include, link, import or execute
// it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// It restates, function by function, the LIVE set of /root/reference/lib/fsearch.py
// (RPython, translated to C by the reference's bin/find_hit.py:198-208).  Every
// function cites the reference lines it follows.  The reference itself cannot be
// compiled here or on the GPU box (needs rpython + python2 from the network), so the
// oracle is PINNED against fixtures produced by running the real reference source
// under CPython in the build container (tools/refharness/, fixtures in
// tests/golden/); tests/test_oracle_golden.py replays them.  Residual unpinned risk:
// the real binary's gcc -ffast-math and RPython's float formatting.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fno-builtin, no fast-math:
// libm log/log10/pow/sqrt must be called exactly as CPython's math module calls them).
//
// Single-threaded on purpose: one process == one reference `fsearch-c` process.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

typedef int64_t i64;

// ---------------------------------------------------------------------------------
// F0  BLOSUM62 as int[256][256] over raw bytes.  fsearch.py:330 (B62 dict, 23 letters
// ABCDEFGHIKLMNPQRSTVWXYZ), 333-346 (dict2mat: both cases of each letter, every other
// byte pair -4).  The 23x23 block below is the standard NCBI BLOSUM62; the golden test
// checks all 65536 entries against the reference's own table.
// ---------------------------------------------------------------------------------
const char B62_LETTERS[] = "ARNDCQEGHILKMFPSTWYVBZX";
const signed char B62_TAB[23][23] = {
    /*A*/ {4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0, -2, -1, 0},
    /*R*/ {-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3, -1, 0, -1},
    /*N*/ {-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3, 3, 0, -1},
    /*D*/ {-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3, 4, 1, -1},
    /*C*/ {0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1, -3, -3, -2},
    /*Q*/ {-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2, 0, 3, -1},
    /*E*/ {-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2, 1, 4, -1},
    /*G*/ {0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3, -1, -2, -1},
    /*H*/ {-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3, 0, 0, -1},
    /*I*/ {-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3, -3, -3, -1},
    /*L*/ {-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1, -4, -3, -1},
    /*K*/ {-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2, 0, 1, -1},
    /*M*/ {-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1, -3, -1, -1},
    /*F*/ {-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1, -3, -3, -1},
    /*P*/ {-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2, -2, -1, -2},
    /*S*/ {1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2, 0, 0, 0},
    /*T*/ {0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0, -1, -1, 0},
    /*W*/ {-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3, -4, -3, -2},
    /*Y*/ {-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1, -3, -2, -1},
    /*V*/ {0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4, -3, -2, -1},
    /*B*/ {-2, -1, 3, 4, -3, 0, 1, -1, 0, -3, -4, 0, -3, -3, -2, 0, -1, -4, -3, -3, 4, 1, -1},
    /*Z*/ {-1, 0, 0, 1, -3, 3, 4, -2, 0, -3, -3, 1, -1, -3, -1, 0, -1, -3, -2, -2, 1, 4, -1},
    /*X*/ {0, -1, -1, -1, -2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -2, 0, 0, -2, -1, -1, -1, -1, -1},
};

int b62[256][256];

void init_b62() {
    static bool done = false;
    if (done) return;
    for (int i = 0; i < 256; ++i)
        for (int j = 0; j < 256; ++j) b62[i][j] = -4;
    for (int a = 0; a < 23; ++a)
        for (int b = 0; b < 23; ++b) {
            int ca[2] = {B62_LETTERS[a], B62_LETTERS[a] + 32};
            int cb[2] = {B62_LETTERS[b], B62_LETTERS[b] + 32};
            for (int x : ca)
                for (int y : cb) b62[x][y] = B62_TAB[a][b];
        }
    done = true;
}

// ---------------------------------------------------------------------------------
// F8  qsort/quicksort/partition/insort (fsearch.py:260-327; *_u twins 189-256 are
// identical).  Non-stable; Rand.init_genrand(42) on every call makes random() ==
// 0.3745401188473625 (first MT19937 res53 after seed 42).  We sort a permutation
// `x` of record indices by `key[x[.]]` -- swapping whole records == swapping indices.
// ---------------------------------------------------------------------------------
const double RAND42 = 0.3745401188473625;

template <class K>
void ref_insort(std::vector<int>& x, int l, int r, const K& key) {  // 266-277
    for (int i = l; i < r; ++i) {
        int v = x[i];
        auto pivot = key(v);
        int j = i - 1;
        while (j >= l) {
            if (key(x[j]) <= pivot) break;
            x[j + 1] = x[j];
            --j;
        }
        x[j + 1] = v;
    }
}

template <class K>
int ref_partition(std::vector<int>& x, int l, int r, const K& key) {  // 281-297
    auto pivot = key(x[l]);
    int i = l, j = r + 1;
    for (;;) {
        ++i;
        while (i <= r && key(x[i]) < pivot) ++i;
        --j;
        while (key(x[j]) > pivot) --j;
        if (i > j) break;
        std::swap(x[i], x[j]);
    }
    std::swap(x[l], x[j]);
    return j;
}

template <class K>
void ref_quicksort(std::vector<int>& x, int l, int r, const K& key) {  // 302-321
    if (r <= l) return;
    int gap = r - l + 1, m;
    if (gap < 7) {
        ref_insort(x, l, r + 1, key);
        return;
    } else if (gap == 7) {
        m = l + gap / 2;
    } else {
        m = l + (int)(RAND42 * gap);
    }
    std::swap(x[l], x[m]);
    int med = ref_partition(x, l, r, key);
    ref_quicksort(x, l, med - 1, key);
    ref_quicksort(x, med + 1, r, key);
}

template <class K>
void ref_qsort(std::vector<int>& x, const K& key) {  // 326-327
    ref_quicksort(x, 0, (int)x.size() - 1, key);
}

// ---------------------------------------------------------------------------------
// F2  generate_nr_tbl (fsearch.py:406-422): 512-entry identity table; every letter of
// a group (both cases) -> smallest ASCII code of the (upper-cased) group.
// ---------------------------------------------------------------------------------
std::vector<int> generate_nr_tbl(const std::string& gaa) {
    std::vector<int> tbl(512);
    for (int i = 0; i < 512; ++i) tbl[i] = i;
    std::string up = gaa;
    for (auto& c : up) c = (char)toupper((unsigned char)c);
    size_t p = 0;
    while (p <= up.size()) {
        size_t q = up.find(',', p);
        if (q == std::string::npos) q = up.size();
        std::string grp = up.substr(p, q - p);
        int flag = 1024;
        for (unsigned char c : grp)
            if (c < flag) flag = c;
        for (unsigned char c : grp) {
            tbl[c] = flag;
            tbl[(unsigned char)tolower(c)] = flag;
        }
        p = q + 1;
    }
    return tbl;
}

std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out;
    size_t p = 0;
    for (;;) {
        size_t q = s.find(sep, p);
        if (q == std::string::npos) {
            out.push_back(s.substr(p));
            break;
        }
        out.push_back(s.substr(p, q - p));
        p = q + 1;
    }
    return out;
}

struct Seed {
    uint32_t bucket;
    int pos;
};

// ---------------------------------------------------------------------------------
// F3  spseeds_fnv (fsearch.py:519-556).  For each alphabet table, each seed pattern s,
// each start i (step): FNV-1a32 over code[c] at care positions, then over the pattern
// index s; window rejected when ANY covered position is x/X; duplicate (bucket, i)
// within one alphabet emitted once.  Emission order: alphabet, pattern, position.
// ---------------------------------------------------------------------------------
void spseeds(const std::string& seq, int step, const std::vector<std::vector<int>>& codes,
             const std::vector<std::string>& spaces, i64 mod, std::vector<Seed>& out) {
    out.clear();
    const i64 L = (i64)seq.size();
    const int S = (int)spaces.size();
    if (step < 1) step = 1;  // reference: xrange step <= 0 raises; never passed
    for (const auto& code : codes) {
        std::unordered_set<uint64_t> visit;
        for (int s = 0; s < S; ++s) {
            const std::string& space = spaces[s];
            const i64 k = (i64)space.size();
            for (i64 i = 0; i < L - k + 1; i += step) {
                bool seg = true;
                uint32_t n = 0x811c9dc5u;
                for (i64 j = 0; j < k; ++j) {
                    unsigned char ch = (unsigned char)seq[i + j];
                    if (ch == 'x' || ch == 'X') {
                        seg = false;
                        break;
                    } else if (space[j] != '0') {
                        n ^= (uint32_t)code[ch];
                        n *= 0x01000193u;
                    }
                }
                n ^= (uint32_t)s;
                n *= 0x01000193u;
                uint32_t nmod = (uint32_t)((i64)n % mod);
                if (!seg) continue;
                if (S > 1) {
                    uint64_t key = ((uint64_t)nmod << 32) | (uint32_t)i;
                    if (!visit.insert(key).second) continue;
                }
                out.push_back({nmod, (int)i});
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// F1  index() / Fasta.__getitem__ (fsearch.py:1543-1553, 2182-2205): record offsets =
// [0] + {i : s[i]=='>' and s[i-1]=='\n'}; hd = first line minus its first byte;
// sq = remaining lines joined (no other stripping).
// ---------------------------------------------------------------------------------
struct Fasta {
    std::string data;
    std::vector<i64> idx;
    i64 N = 0;
    bool load(const char* path) {
        FILE* f = fopen(path, "rb");
        if (!f) return false;
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        data.resize((size_t)n);
        if (n > 0 && fread(&data[0], 1, (size_t)n, f) != (size_t)n) {
            fclose(f);
            return false;
        }
        fclose(f);
        scan();
        return true;
    }
    void from_bytes(const char* p, i64 n) {
        data.assign(p, (size_t)n);
        scan();
    }
    void scan() {
        idx.clear();
        idx.push_back(0);
        const i64 n = (i64)data.size();
        for (i64 i = 1; i < n; ++i)
            if (data[i] == '>' && data[i - 1] == '\n') idx.push_back(i);
        N = (i64)idx.size();
    }
    void get(i64 x, std::string& hd, std::string& sq) const {
        hd.clear();
        sq.clear();
        if (x < 0) x += N;
        if (x < 0 || x >= N) return;
        i64 st = idx[x], ed = (x == N - 1) ? (i64)data.size() : idx[x + 1];
        i64 p = st;
        while (p < ed && data[p] != '\n') ++p;
        if (p > st) hd.assign(data, st + 1, p - st - 1);
        ++p;
        while (p < ed) {
            i64 q = p;
            while (q < ed && data[q] != '\n') ++q;
            sq.append(data, p, q - p);
            p = q + 1;
        }
    }
};

// ---------------------------------------------------------------------------------
// F6  seg / entropy / Counter (fsearch.py:2872-2928, 2854-2868, 157-177).  Only
// output[:n] (first return value) is used by blastp (2996, 3034).
// ---------------------------------------------------------------------------------
std::string seg(const std::string& S) {
    const double minent = 2.2, window = 12.;
    std::string s = S;
    for (auto& c : s) c = (char)toupper((unsigned char)c);
    const double log2v = log(2);
    const i64 n = (i64)s.size();
    const int winsize = 12;
    if (n == 0) return s;  // reference: IndexError on mask[0]; defined here as a no-op
    // entropy(s[:12]) with the Counter quirk: value = 2*occ - 1, values() in first-seen order
    double counts[256];
    bool seen[256];
    for (int i = 0; i < 256; ++i) counts[i] = 0, seen[i] = false;
    std::vector<int> order;
    const i64 w = std::min<i64>(n, winsize);
    for (i64 i = 0; i < w; ++i) {
        unsigned char c = (unsigned char)s[i];
        if (seen[c]) counts[c] += 1;
        else seen[c] = true, counts[c] = 0, order.push_back(c);
    }
    for (i64 i = 0; i < w; ++i) counts[(unsigned char)s[i]] += 1.;
    double ent = 0;
    const double nn = (double)w * 1.;
    for (int c : order) {
        double freq = counts[c] / nn;
        ent -= freq * log(freq);
    }
    ent /= log(2);
    std::vector<char> mask((size_t)n, 0);
    if (ent < minent) mask[0] = 1;
    for (i64 i = 1; i < n - winsize + 1; ++i) {
        unsigned char pre = (unsigned char)s[i - 1], cur = (unsigned char)s[i + 11];
        if (pre == cur) {
            mask[i] = mask[i - 1];
            continue;
        }
        double pre_count = counts[pre];
        counts[pre] -= 1;
        double cur_count = counts[cur];
        counts[cur] += 1;
        double a = pre_count / window, b = counts[pre] / window;
        // (b != 0 and X or Y): Y also when X == 0.0
        double t;
        if (b != 0) {
            t = (a * log(a) - b * log(b)) / log2v;
            if (t == 0) t = a * log(a) / log2v;
        } else {
            t = a * log(a) / log2v;
        }
        ent += t;
        a = cur_count / window;
        b = counts[cur] / window;
        if (a != 0) {
            t = (a * log(a) - b * log(b)) / log2v;
            if (t == 0) t = -b * log(b) / log2v;
        } else {
            t = -b * log(b) / log2v;
        }
        ent += t;
        if (ent < minent) mask[i] = 1;
    }
    i64 Nws = std::max<i64>(0, n - winsize);
    if (mask[Nws] == 1)
        for (i64 i = Nws; i < n; ++i) mask[i] = 1;
    std::string output;
    output.reserve((size_t)n + 12);
    i64 st = 0;
    while (st < n) {
        if (mask[st] == 0) {
            output.push_back(s[st]);
            st += 1;
        } else {
            output.append(12, 'x');
            st += 12;
        }
    }
    output.resize((size_t)n);
    return output;
}

// ---------------------------------------------------------------------------------
// F14  score2bit (fsearch.py:1066-1071), bit2e (1086)
// ---------------------------------------------------------------------------------
i64 score2bit(i64 score) {
    double bit = (.267 * (double)score + 3.1941832122778293) / 0.69314718055994529;
    return (i64)bit;
}

double bit2e(i64 D, i64 li, i64 lj, i64 bit) { return (double)(D * li * lj) * pow(2, (double)(-bit)); }

// ---------------------------------------------------------------------------------
// F15  f2s (fsearch.py:43-61) with RPython str(float) == '%f'
// ---------------------------------------------------------------------------------
std::string fmt_f(double x) {
    char buf[512];
    snprintf(buf, sizeof buf, "%f", x);
    return buf;
}

std::string f2s(double e) {
    if (e <= 0) return "0";
    if (e < 1e-3) {
        double a = log10(e);
        a -= (double)(i64)a;
        if (a < 0) {
            double t = 1 + a;
            a = (t != 0) ? t : a;  // `a < 0 and 1 + a or a`
        }
        double b = pow(10, a);
        std::string s = fmt_f(log10(e / b));
        size_t d = s.find('.');
        s = s.substr(0, d == std::string::npos ? 0 : d);
        std::string p = fmt_f(b);
        d = p.find('.');
        p = p.substr(0, d == std::string::npos ? 2 : d + 3);  // max(0, find+3); find==-1 -> 2
        return p + "e" + s;
    }
    return fmt_f(e);
}

std::string fmt_idy(double idy) {  // entry_point 3235-3237
    std::string s = fmt_f(idy);
    size_t d = s.find('.');
    return s.substr(0, d == std::string::npos ? 2 : d + 3);
}

// ---------------------------------------------------------------------------------
// F13  kswat_st (fsearch.py:1357-1476) on the persistent 4100x4100 score/trace
// matrices of blastp (2982-2983) -- never cleared between calls, exactly as there.
// ---------------------------------------------------------------------------------
const int MATN = 4100;
struct Mats {
    std::vector<int> score;
    std::vector<char> trace;
    Mats() : score((size_t)MATN * MATN, 0), trace((size_t)MATN * MATN, '*') {}
    int& S(i64 i, i64 j) { return score[(size_t)i * MATN + (size_t)j]; }
    char& T(i64 i, i64 j) { return trace[(size_t)i * MATN + (size_t)j]; }
};

struct Aln {
    double idy;
    i64 aln, mis, gap, qst, qed, sst, sed, bit;
    i64 cells;  // DP cells evaluated (work counter, not in the reference)
    i64 maxscore;
};

Aln kswat_st(const std::string& S0, const std::string& S1, i64 qst, i64 sst, Mats& M) {
    const int go = -11, ge = -1, kbound = 16;
    i64 qed = -1, sed = -1;
    qst = std::min<i64>(std::max<i64>(qst, 0), (i64)S0.size());
    qed = (i64)S0.size();
    sst = std::min<i64>(std::max<i64>(sst, 0), (i64)S1.size());
    sed = (i64)S1.size();
    const std::string *s0, *s1;
    bool swap;
    if (std::llabs(qed - qst) < std::llabs(sed - sst)) {
        s0 = &S0, s1 = &S1, swap = false;
    } else {
        s0 = &S1, s1 = &S0, swap = true;
        std::swap(qst, sst);
        std::swap(qed, sed);
    }
    const i64 qsp = qst < qed ? 1 : -1, ssp = sst < sed ? 1 : -1;
    const i64 l0 = std::llabs(qed - qst) + 1, l1 = std::llabs(sed - sst) + 1;
    for (i64 i = 1; i < l0; ++i) M.S(0, i) = 0, M.T(0, i) = '-';
    for (i64 i = 1; i < l1; ++i) {
        M.S(i, 0) = 0;
        M.T(i, 0) = '|';
        i64 start = std::max<i64>(0, i - kbound - 1), end = std::min<i64>(i + kbound + 1, l0 - 1);
        M.T(i, start) = '|';
        M.T(i, end) = '-';
        M.S(i, start) = 0;
        M.S(i, end) = 0;
    }
    i64 i_max = 0, j_max = 0, cells = 0;
    int maxscore = 0;
    for (i64 i = 1; i < l1; ++i) {
        i64 start = std::max<i64>(1, i - kbound), end = std::min<i64>(i + kbound, l0);
        for (i64 j = start; j < end; ++j) {
            int I = M.S(i, j - 1) + (M.T(i, j - 1) == '-' ? ge : go);
            unsigned char c1 = (unsigned char)(*s1)[(i - 1) * ssp + sst], c0 = (unsigned char)(*s0)[(j - 1) * qsp + qst];
            int Mv = M.S(i - 1, j - 1) + b62[c1][c0];
            int D = M.S(i - 1, j) + (M.T(i - 1, j) == '|' ? ge : go);
            int B = std::max(std::max(0, I), std::max(Mv, D));
            M.S(i, j) = B;
            ++cells;
            if (B > maxscore) i_max = i, j_max = j, maxscore = B;
            if (B == Mv) M.T(i, j) = '\\';
            else if (B == I) M.T(i, j) = '-';
            else if (B == D) M.T(i, j) = '|';
            else M.T(i, j) = '*';
        }
    }
    i64 i = i_max, j = j_max;
    std::vector<int> al0, al1;  // byte or -1 for '-'... '-' is a byte too: keep raw bytes
    while (i > 0 || j > 0) {
        char t = M.T(i, j);
        if (t == '\\') {
            al0.push_back((unsigned char)(*s0)[(j - 1) * qsp + qst]);
            al1.push_back((unsigned char)(*s1)[(i - 1) * ssp + sst]);
            --i, --j;
        } else if (t == '-') {
            al0.push_back((unsigned char)(*s0)[(j - 1) * qsp + qst]);
            al1.push_back('-');
            --j;
        } else if (t == '|') {
            al1.push_back((unsigned char)(*s1)[(i - 1) * ssp + sst]);
            al0.push_back('-');
            --i;
        } else {
            break;
        }
    }
    if (qst < qed) std::reverse(al0.begin(), al0.end());
    else std::swap(i, i_max);
    if (sst < sed) std::reverse(al1.begin(), al1.end());
    else std::swap(j, j_max);
    const i64 AL = (i64)al0.size();
    double idy = 0;
    i64 mis = 0, gap = 0;
    int op = -1;
    for (i64 k = 0; k < AL; ++k) {
        if (al0[k] == al1[k]) idy += 1.;
        else mis += 1;
        if (al0[k] == '-' && op != 0) gap += 1, op = 0;
        else if (al1[k] == '-' && op != 1) gap += 1, op = 1;
        else op = -1;
    }
    idy *= (100. / (double)AL);  // AL == 0 -> nan (RPython C semantics; CPython would raise)
    Aln r;
    r.idy = idy, r.aln = AL, r.mis = mis, r.gap = gap, r.cells = cells, r.maxscore = maxscore;
    r.bit = score2bit(maxscore);
    if (swap) {
        r.qst = i * ssp + sst, r.qed = i_max * ssp + sst, r.sst = j * qsp + qst, r.sed = j_max * qsp + qst;
    } else {
        r.qst = j * qsp + qst, r.qed = j_max * qsp + qst, r.sst = i * qsp + sst, r.sed = i_max * qsp + sst;
    }
    return r;
}

// kswat_st_long (fsearch.py:1480-1498): independent 4096x4096 tiles along the diagonal
void kswat_st_long(const std::string& sqi, const std::string& sqj, i64 qi, i64 qj, Mats& M, std::vector<Aln>& out) {
    const i64 chk = 4096, li = (i64)sqi.size();
    i64 j = qj;
    out.clear();
    for (i64 i0 = qi; i0 < li; i0 += chk) {
        i64 i = std::max<i64>(0, i0), ied = std::max<i64>(0, i0 + chk);
        j = std::max<i64>(0, j);
        i64 jed = std::max<i64>(0, j + chk);
        std::string a = i < (i64)sqi.size() ? sqi.substr((size_t)i, (size_t)(ied - i)) : std::string();
        std::string b = j < (i64)sqj.size() ? sqj.substr((size_t)j, (size_t)(jed - j)) : std::string();
        Aln r = kswat_st(a, b, 0, 0, M);
        r.qst += i, r.qed += i, r.sst += j, r.sed += j;
        out.push_back(r);
        j += chk;
    }
}

// ---------------------------------------------------------------------------------
// F10  Fasta.ungap (fsearch.py:2454-2494) and get_ungap_scores (2497-2509)
// ---------------------------------------------------------------------------------
struct Ungap {
    i64 max_score, max_qst, max_qed, max_sst, max_sed, flag;
};

Ungap ungap(const std::string& qseq, const std::string& sseq, i64 Qst, i64 Sst, i64 qlo = -1, i64 slo = -1) {
    const int dropX = 30;
    qlo = qlo > -1 ? qlo : 0;  // `qlo > -1 and qlo or 0`
    slo = slo > -1 ? slo : 0;
    const i64 ql = (i64)qseq.size(), sl = (i64)sseq.size();
    const i64 qup = ql, sup = sl;
    i64 off = std::max<i64>(std::max<i64>(qlo - Qst, slo - Sst), 0);
    Qst += off;
    Sst += off;
    i64 qst = Qst, sst = Sst;
    i64 score = 0, max_score = 0, max_qed = qst, max_sed = sst, flag = 0;
    while (qlo < qst && qst < qup && slo < sst && sst < sup) {
        ++flag;
        score += b62[(unsigned char)qseq[qst]][(unsigned char)sseq[sst]];
        if (score > max_score) max_score = score, max_qed = qst, max_sed = sst;
        else if (score + dropX < max_score) break;
        ++qst, ++sst;
    }
    qst = Qst - 1, sst = Sst - 1;
    score = max_score;
    i64 max_qst = qst, max_sst = sst;
    while (qup > qst && qst > qlo && sup > sst && sst > slo) {
        ++flag;
        score += b62[(unsigned char)qseq[qst]][(unsigned char)sseq[sst]];
        if (score > max_score) max_score = score, max_qst = qst, max_sst = sst;
        else if (score + dropX < max_score) break;
        --qst, --sst;
    }
    return {max_score, max_qst, max_qed, max_sst, max_sed, flag};
}

struct UngapChain {
    i64 score, flag, x0, y0, x, y;
};

UngapChain get_ungap_scores(const std::string& qseq, const std::string& sseq, const std::vector<std::pair<int, int>>& loc1) {
    Ungap u = ungap(qseq, sseq, loc1[0].first, loc1[0].second);
    i64 scores = u.max_score, flag = u.flag;
    i64 x0 = u.max_qst, y0 = u.max_sst, x = u.max_qed, y = u.max_sed;
    for (size_t k = 1; k < loc1.size(); ++k) {
        Ungap v = ungap(qseq, sseq, loc1[k].first, loc1[k].second, x, y);
        flag += v.flag;
        x = v.max_qed, y = v.max_sed;
        scores += v.max_score;
    }
    return {scores, flag, x0, y0, x, y};
}

// ---------------------------------------------------------------------------------
// F9  lis (fsearch.py:688-724) keyed by sst over the qst-sorted group
// ---------------------------------------------------------------------------------
std::vector<std::pair<int, int>> lis(const std::vector<std::pair<int, int>>& seq) {
    const int N = (int)seq.size();
    if (N < 2) return seq;
    std::vector<int> Mv(N, -1), P(N, -1);
    int L = 1;
    Mv[0] = 0;
    auto key = [&](int i) { return seq[i].second; };
    for (int i = 1; i < N; ++i) {
        int lower = 0, upper = L, j;
        if (key(Mv[upper - 1]) < key(i)) {
            j = upper;
        } else {
            while (upper - lower > 1) {
                int mid = (upper + lower) / 2;
                if (key(Mv[mid - 1]) < key(i)) lower = mid;
                else upper = mid;
            }
            j = lower;
        }
        P[i] = j - 1 >= 0 ? Mv[j - 1] : Mv[N - 1];  // python M[-1] wraps to the last slot
        if (j == L || key(i) < key(Mv[j])) {
            Mv[j] = i;
            L = std::max(L, j + 1);
        }
    }
    std::vector<std::pair<int, int>> result;
    int pos = Mv[L - 1];
    for (int e = 0; e < L; ++e) {
        result.push_back(seq[pos < 0 ? pos + N : pos]);
        pos = P[pos < 0 ? pos + N : pos];
    }
    std::reverse(result.begin(), result.end());
    return result;
}

// ---------------------------------------------------------------------------------
// F4  Fasta.build_msav (fsearch.py:2208-2280) + get_mu_sd (746-761)
// ---------------------------------------------------------------------------------
struct Params {
    std::string ssd = "111111";
    std::string nr = "AST,CFILMVY,DN,EQ,G,H,KR,P,W";
    double expect = 1e-3, max_miss = 1e-3;
    i64 v = 500, st = -1, ed = -1, rst = -1, red = -1, thr = -1, step = 4, ht = -1, chk = 50000;
    std::string flt = "T";
    std::string tmpdir = "./tmpdir";
};

struct Cand {
    uint32_t subj, score, qi, qj;
};

// NC = ht < 1 and bins or ht (fsearch.py:2228-2231).  self.scale is overwritten with the parameter (-1, line 2216), so
// bins = min(int(pow(-1, mw)) * nssp * 5, 128Mi): 5 * nssp for an even maximum seed weight; negative for an odd one
// (an empty table and an IndexError in the reference) -> returned as is, callers refuse values < 1.
i64 resolve_nc(const Params& p) {
    if (p.ht >= 1) return p.ht;
    int mw = 0;
    for (auto& s : split(p.ssd, ',')) mw = std::max(mw, (int)std::count(s.begin(), s.end(), '1'));
    const i64 nssp = (i64)std::count(p.ssd.begin(), p.ssd.end(), ',') + 1;
    return std::min<i64>((mw % 2 == 0 ? 1 : -1) * nssp * 5, 128ll * 1024 * 1024);
}

struct Index {
    std::vector<std::vector<int>> codes;
    std::vector<std::string> spaces;
    int mink = 0;
    i64 NC = 0;
    std::vector<uint32_t> start, locus, soas;
    i64 threshold = 0, L = -1, offset = 0, offend = 0;
    std::vector<std::pair<std::string, std::string>> hdseqs;
    // work counters (not in the reference)
    i64 n_seed_hits = 0, n_groups = 0, n_ungap_steps = 0;

    void build(const Fasta& fa, const Params& p, i64 start_, i64 end_) {
        init_b62();
        codes.clear();
        for (auto& e : split(p.nr, '/')) codes.push_back(generate_nr_tbl(e));
        spaces = split(p.ssd, ',');
        mink = 1 << 30;
        for (auto& s : spaces) mink = std::min(mink, (int)s.size());
        offset = start_;
        offend = end_ + 1;
        NC = resolve_nc(p);
        if (NC < 1) NC = 0;  // refused by the callers (blastp / oc_index_build)
        start.assign((size_t)NC, 0);
        i64 st = std::min<i64>(std::max<i64>(0, start_), fa.N);
        i64 ed = std::min<i64>(end_ < 0 ? fa.N : end_, fa.N);
        i64 Mn = ed - st;
        if (Mn < 0) Mn = 0;
        soas.assign((size_t)Mn + 1, 0);
        std::string hd, sq;
        std::vector<Seed> seeds;
        for (i64 i = st; i < ed; ++i) {
            fa.get(i, hd, sq);
            i64 j = i - st;
            soas[j + 1] = (uint32_t)((i64)soas[j] + (i64)sq.size());
            spseeds(sq, (int)p.step, codes, spaces, NC, seeds);
            for (auto& s : seeds) start[s.bucket] += 1;
        }
        // get_mu_sd(self.start), m = 0
        {
            i64 Nn = 1;
            double mu = 0.;
            for (i64 b = 0; b < NC; ++b) {
                i64 c = start[b];
                if (c > 0) mu += (double)c, Nn += 1;
            }
            mu /= (double)Nn;
            double sd = 0.;
            for (i64 b = 0; b < NC; ++b) {
                i64 c = start[b];
                if (c > 0) sd += pow((double)c - mu, 2);
            }
            sd = sqrt(sd / (double)Nn);
            threshold = (i64)(mu + 2 * sd);
        }
        for (i64 b = 1; b < NC; ++b) start[b] = (uint32_t)((i64)start[b - 1] + (i64)start[b]);
        locus.assign(NC > 0 ? (size_t)start[NC - 1] : 0, 0);
        for (i64 i = st; i < ed; ++i) {
            fa.get(i, hd, sq);
            i64 off = soas[i - st];
            spseeds(sq, (int)p.step, codes, spaces, NC, seeds);
            for (auto& s : seeds) {
                start[s.bucket] -= 1;
                locus[start[s.bucket]] = (uint32_t)(s.pos + off);
            }
        }
        L = (i64)locus.size() - 1;
        hdseqs.clear();
        for (i64 e = offset; e < offend; ++e) {
            fa.get(e, hd, sq);  // out of range -> ['', '']
            if (e < 0 || e >= fa.N) hd.clear(), sq.clear();
            hdseqs.emplace_back(hd, sq);
        }
    }

    // get_bin_mem (fsearch.py:2530-2541)
    void get_bin(i64 i, i64& st, i64& ed) const {
        i = i > 0 ? i : 0;
        i64 a = start[i], b = (i + 1 < NC) ? (i64)start[i + 1] : a;
        st = std::max<i64>(a, 0);
        ed = std::min<i64>(std::max<i64>(b, 0), L);
    }

    // bisect(self.soas, x) (fsearch.py:134-153): `l = l < 0 and 0 or l` leaves l = -1
    i64 bisect_soas(i64 x) const {
        i64 l = -1, r = (i64)soas.size();
        while (r - l > 1) {
            i64 m = (l + r) / 2;
            if ((i64)soas[m] < x) l = m;
            else r = m;
        }
        return l;
    }

    // -----------------------------------------------------------------------------
    // F7 + F11  Fasta.find_msav_m (fsearch.py:2645-2724), guess_start (2544-2553)
    // `tuples` (optional) receives every seed hit in visiting order as
    // (subject, diag = qst - sst, qst); `marks` the cap-selected query positions.
    // -----------------------------------------------------------------------------
    void find_msav_m(const std::string& seq, std::vector<Cand>& out, std::vector<int32_t>* tuples = nullptr,
                     std::vector<char>* marks = nullptr) {
        out.clear();
        const i64 ql = (i64)seq.size();
        if (ql < mink) return;  // reference indexes past the string (undefined); defined here: no hits
        const i64 nk = ql - mink + 1;
        std::vector<int> kscs((size_t)nk, 0);
        int sc = 0;
        for (int i = 0; i < mink; ++i) {
            unsigned char c = (unsigned char)seq[i];
            sc += b62[c][c];
        }
        kscs[0] = sc;
        for (i64 i = 1; i < nk; ++i) {
            unsigned char c0 = (unsigned char)seq[i - 1], c1 = (unsigned char)seq[i - 1 + mink];
            sc = kscs[i - 1] - b62[c0][c0] + b62[c1][c1];
            kscs[i] = sc;
        }
        std::vector<Seed> s2a;
        spseeds(seq, 1, codes, spaces, NC, s2a);
        std::vector<i64> cnt((size_t)nk, 0);
        for (auto& s : s2a) {
            i64 st, ed;
            get_bin(s.bucket, st, ed);
            i64 c = ed - st;
            cnt[s.pos] += c > 0 ? c : 0;
        }
        const i64 thr = threshold * ql;
        std::vector<int> hist((size_t)nk);
        for (i64 i = 0; i < nk; ++i) hist[i] = (int)i;
        ref_qsort(hist, [&](int i) { return -kscs[i]; });
        std::vector<char> hist_c((size_t)ql, 0);
        i64 cum = 0;
        for (i64 i = 0; i < nk; ++i) {
            if (cum > thr) break;
            cum += cnt[hist[i]];
            hist_c[hist[i]] = 1;
        }
        if (marks) *marks = hist_c;

        // hits dict keyed (hd, k0) in insertion order
        struct Group {
            i64 hd, k0;
            std::vector<std::pair<int, int>> loc;
        };
        std::vector<Group> groups;
        std::unordered_map<uint64_t, int> gmap;
        for (auto& s : s2a) {
            if (!hist_c[s.pos]) continue;
            i64 st, ed;
            get_bin(s.bucket, st, ed);
            for (i64 slot = st; slot < ed; ++slot) {
                i64 x = locus[slot];
                i64 idx = bisect_soas(x);
                i64 hd = idx + offset;
                i64 sst = x - (i64)soas[idx < 0 ? idx + (i64)soas.size() : idx];
                i64 k0 = s.pos - sst;
                ++n_seed_hits;
                if (tuples) tuples->push_back((int32_t)hd), tuples->push_back((int32_t)k0), tuples->push_back((int32_t)s.pos);
                uint64_t key = ((uint64_t)(uint32_t)(int32_t)hd << 32) | (uint32_t)(int32_t)k0;
                auto it = gmap.find(key);
                int g;
                if (it == gmap.end()) {
                    g = (int)groups.size();
                    gmap.emplace(key, g);
                    groups.push_back({hd, k0, {}});
                } else {
                    g = it->second;
                }
                groups[g].loc.emplace_back(s.pos, (int)sst);
            }
        }
        n_groups += (i64)groups.size();

        // per-group chained ungapped X-drop; best diagonal per subject, first wins ties
        struct Best {
            i64 hd, score, qst, sst, qed, sed;
        };
        std::vector<Best> best;
        std::unordered_map<i64, int> bmap;
        for (auto& g : groups) {
            i64 k = g.hd - offset;
            if (k < 0) k += (i64)hdseqs.size();
            const std::string& sseq = hdseqs[(size_t)k].second;
            std::vector<int> perm(g.loc.size());
            for (size_t i = 0; i < perm.size(); ++i) perm[i] = (int)i;
            ref_qsort(perm, [&](int i) { return g.loc[i].first; });
            std::vector<std::pair<int, int>> loc0(perm.size());
            for (size_t i = 0; i < perm.size(); ++i) loc0[i] = g.loc[perm[i]];
            std::vector<std::pair<int, int>> loc1 = lis(loc0);
            UngapChain u = get_ungap_scores(seq, sseq, loc1);
            n_ungap_steps += u.flag;
            if (u.score < 25) continue;
            auto it = bmap.find(g.hd);
            if (it == bmap.end()) {
                bmap.emplace(g.hd, (int)best.size());
                best.push_back({g.hd, u.score, u.x0, u.y0, u.x, u.y});
            } else if (u.score > best[it->second].score) {
                Best& b = best[it->second];
                b.score = u.score, b.qst = u.x0, b.sst = u.y0, b.qed = u.x, b.sed = u.y;
            }
        }
        for (auto& b : best) {
            // guess_start over [[qst, sst], [qed, sed]]: dist = floor(sum(sst - qst) / 2)
            i64 dist = (b.sst - b.qst) + (b.sed - b.qed);
            dist = (dist >= 0) ? dist / 2 : -((-dist + 1) / 2);
            i64 qi, qj;
            if (dist > 0) qi = 0, qj = dist;
            else qi = -dist, qj = 0;
            out.push_back({(uint32_t)b.hd, (uint32_t)b.score, (uint32_t)qi, (uint32_t)qj});
        }
    }
};

// ---------------------------------------------------------------------------------
// F12  blastp (fsearch.py:2968-3121) + F15 row formatting (entry_point 3231-3258)
// ---------------------------------------------------------------------------------
struct Row {
    i64 qidx, sidx;
    std::string text;
};

struct Stats {
    i64 n_queries = 0, query_aa = 0, rows = 0, seed_hits = 0, groups = 0, ungap_steps = 0, cands = 0, alignments = 0,
        cells = 0;
    double t_index = 0, t_seed = 0, t_align = 0;
};

double now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

struct HitRec {  // one reported alignment (fixed-width; mirrors include/sohit.h's so_hit)
    i64 qidx, sidx;
    double idy, e;
    i64 aln, mis, gap, qst, qed, sst, sed, bit, qlen, slen, ungapped;
};

int blastp(const char* qry, const char* ref, const Params& p0, FILE* out, Stats& stt, std::vector<HitRec>* recs = nullptr,
           std::vector<std::vector<Cand>>* cand_dump = nullptr) {
    init_b62();
    Params p = p0;
    p.max_miss = std::max(p.max_miss, 1e-3);
    if (resolve_nc(p) < 1) {
        fprintf(stderr, "sohit_cpu: -M < 1 with an odd seed weight gives the reference a negative table size (fsearch.py:2228-2231)\n");
        return 2;
    }
    Fasta seqs, DB;
    if (!seqs.load(qry) || !DB.load(ref)) return 1;
    const i64 N = seqs.N, D = DB.N;
    i64 st = std::min<i64>(std::max<i64>(0, p.st), N);
    i64 ed = std::min<i64>(p.ed < 0 ? D : p.ed, N);
    static thread_local Mats* mats = nullptr;   // (per thread: oracle.blastp_parallel runs query ranges side by side)
    if (!mats) mats = new Mats();
    std::vector<std::vector<Cand>> KDB((size_t)std::max<i64>(0, ed - st));
    std::vector<std::string> masked((size_t)std::max<i64>(0, ed - st));
    std::string hd, sq, hdj, sqj;
    for (i64 i = st; i < ed; ++i) {
        seqs.get(i, hd, sq);
        masked[i - st] = (p.flt == "T") ? seg(sq) : sq;
        stt.query_aa += (i64)sq.size();
    }
    stt.n_queries += std::max<i64>(0, ed - st);
    // PHASE 1 (2990-3020): per reference chunk, build the index and collect candidates
    Index ix;
    i64 Start = p.rst == -1 ? 0 : std::max<i64>(0, p.rst);
    i64 End = p.red == -1 ? D : p.red;
    std::vector<Cand> hits;
    for (i64 c = Start; c < End; c += p.chk) {
        double t0 = now();
        ix.build(DB, p, c, std::min<i64>(c + p.chk, End));
        // `DB.threshold = thr < 1 and DB.threshold or thr` (2992): a zero threshold falls through to thr
        if (p.thr >= 1 || ix.threshold == 0) ix.threshold = p.thr;
        double t1 = now();
        stt.t_index += t1 - t0;
        for (i64 i = st; i < ed; ++i) {
            ix.find_msav_m(masked[i - st], hits);
            auto& k = KDB[i - st];
            k.insert(k.end(), hits.begin(), hits.end());
        }
        stt.t_seed += now() - t1;
    }
    stt.seed_hits += ix.n_seed_hits, stt.groups += ix.n_groups, stt.ungap_steps += ix.n_ungap_steps;
    if (cand_dump) *cand_dump = KDB;
    // PHASE 2 (3028-3110)
    double t2 = now();
    std::vector<Aln> longres;
    for (i64 i = st; i < ed; ++i) {
        auto& H = KDB[i - st];
        stt.cands += (i64)H.size();
        const std::string& sqi = masked[i - st];
        seqs.get(i, hd, sq);
        const i64 li = (i64)sqi.size();
        std::vector<int> perm(H.size());
        for (size_t k = 0; k < perm.size(); ++k) perm[k] = (int)k;
        ref_qsort(perm, [&](int k) { return -(i64)H[k].score; });
        double mmiss = (double)H.size() * p.max_miss + 1;
        mmiss = std::max(mmiss, 100. / mmiss);
        mmiss = std::min(std::max(mmiss, 10.), 120.);
        i64 unmch = 0, bv = 0;
        // vmax = max(100, max(v + 100, v * 1.1)) used as a slice bound (int under RPython)
        i64 vmax = std::max<i64>(100, std::max<i64>(p.v + 100, (i64)((double)p.v * 1.1)));
        struct M8 {
            HitRec r;
            std::string hj, desc;
        };
        std::vector<M8> m8s;
        std::string hi = hd.substr(0, hd.find(' '));
        const i64 lim = std::min<i64>((i64)perm.size(), vmax);
        for (i64 h = 0; h < lim; ++h) {
            const Cand& c = H[perm[h]];
            i64 j = c.subj;
            DB.get(j, hdj, sqj);
            const i64 lj = (i64)sqj.size();
            auto emit = [&](const Aln& a, double e) {
                M8 m;
                m.r = {i, j, a.idy, e, a.aln, a.mis, a.gap, a.qst + 1, a.qed, a.sst + 1, a.sed, a.bit, li, lj, (i64)c.score};
                m.hj = hdj.substr(0, hdj.find(' '));
                m.desc = hdj;
                m8s.push_back(m);
            };
            if (li < 4096 && lj < 4096) {
                Aln a = kswat_st(sqi, sqj, c.qi, c.qj, *mats);
                stt.alignments += 1, stt.cells += a.cells;
                double e = bit2e(D, li, lj, a.bit);
                if (e <= p.expect) {
                    emit(a, e);
                    unmch = 0;
                    bv += 1;
                } else {
                    unmch += 1;
                }
            } else {
                int flag = 1;
                kswat_st_long(sqi, sqj, c.qi, c.qj, *mats, longres);
                for (auto& a : longres) {
                    stt.alignments += 1, stt.cells += a.cells;
                    double e = bit2e(D, li, lj, a.bit);
                    if (e <= p.expect) {
                        emit(a, e);
                        flag = 0;
                        bv += 1;
                    }
                }
                if (flag == 1) unmch += 1;
                else unmch = 0;
            }
            if ((double)unmch >= mmiss || (double)bv >= (double)p.v + mmiss) break;
        }
        std::vector<int> pm(m8s.size());
        for (size_t k = 0; k < pm.size(); ++k) pm[k] = (int)k;
        ref_qsort(pm, [&](int k) { return -m8s[k].r.bit; });
        const i64 nout = std::min<i64>((i64)pm.size(), std::max<i64>(0, p.v));
        for (i64 k = 0; k < nout; ++k) {
            const M8& m = m8s[pm[k]];
            if (!(m.r.e <= p.expect)) continue;
            stt.rows += 1;
            if (recs) recs->push_back(m.r);
            if (out) {
                fprintf(out, "%s\t%s\t%s\t%lld\t%lld\t%lld\t%lld\t%lld\t%lld\t%lld\t%s\t%lld\t%lld\t%lld\t%lld\t%s\n", hi.c_str(),
                        m.hj.c_str(), fmt_idy(m.r.idy).c_str(), (long long)m.r.aln, (long long)m.r.mis, (long long)m.r.gap,
                        (long long)m.r.qst, (long long)m.r.qed, (long long)m.r.sst, (long long)m.r.sed, f2s(m.r.e).c_str(),
                        (long long)m.r.bit, (long long)m.r.qlen, (long long)m.r.slen, (long long)m.r.qidx, m.desc.c_str());
            }
        }
    }
    stt.t_align += now() - t2;
    return 0;
}

}  // namespace

// =================================================================================
// C entry points used by tests/ (ctypes) -- names oc_*
// =================================================================================
extern "C" {

int oc_b62(int a, int b) {
    init_b62();
    return b62[a & 255][b & 255];
}

void oc_nr_tbl(const char* gaa, int* out512) {
    auto t = generate_nr_tbl(gaa);
    for (int i = 0; i < 512; ++i) out512[i] = t[i];
}

// returns number of seeds; buckets/pos sized >= n_alpha * n_patterns * len
i64 oc_spseeds(const char* seq, i64 n, int step, const char* nr, const char* ssd, i64 mod, uint32_t* buckets, int32_t* pos) {
    std::vector<std::vector<int>> codes;
    for (auto& e : split(nr, '/')) codes.push_back(generate_nr_tbl(e));
    std::vector<Seed> out;
    spseeds(std::string(seq, (size_t)n), step, codes, split(ssd, ','), mod, out);
    for (size_t i = 0; i < out.size(); ++i) buckets[i] = out[i].bucket, pos[i] = out[i].pos;
    return (i64)out.size();
}

void oc_seg(const char* s, i64 n, char* out) {
    std::string r = seg(std::string(s, (size_t)n));
    memcpy(out, r.data(), r.size());
}

// sort indices 0..n-1 by key[] with the reference quicksort; perm out
void oc_qsort(const i64* key, int n, int* perm) {
    std::vector<int> x((size_t)n);
    for (int i = 0; i < n; ++i) x[i] = i;
    ref_qsort(x, [&](int i) { return key[i]; });
    for (int i = 0; i < n; ++i) perm[i] = x[i];
}

void oc_ungap(const char* q, i64 ql, const char* s, i64 sl, i64 Qst, i64 Sst, i64 qlo, i64 slo, i64* out6) {
    init_b62();
    Ungap u = ungap(std::string(q, (size_t)ql), std::string(s, (size_t)sl), Qst, Sst, qlo, slo);
    out6[0] = u.max_score, out6[1] = u.max_qst, out6[2] = u.max_qed, out6[3] = u.max_sst, out6[4] = u.max_sed, out6[5] = u.flag;
}

void oc_ungap_chain(const char* q, i64 ql, const char* s, i64 sl, const int32_t* locs, int nloc, i64* out6) {
    init_b62();
    std::vector<std::pair<int, int>> l;
    for (int i = 0; i < nloc; ++i) l.emplace_back(locs[2 * i], locs[2 * i + 1]);
    UngapChain u = get_ungap_scores(std::string(q, (size_t)ql), std::string(s, (size_t)sl), l);
    out6[0] = u.score, out6[1] = u.flag, out6[2] = u.x0, out6[3] = u.y0, out6[4] = u.x, out6[5] = u.y;
}

static Mats* g_mats = nullptr;

// out9: idy(as double bits via *idy), aln, mis, gap, qst, qed, sst, sed, bit ; out[9]=cells, out[10]=maxscore
void oc_kswat_st(const char* q, i64 ql, const char* s, i64 sl, i64 qst, i64 sst, double* idy, i64* out) {
    init_b62();
    if (!g_mats) g_mats = new Mats();
    Aln a = kswat_st(std::string(q, (size_t)ql), std::string(s, (size_t)sl), qst, sst, *g_mats);
    *idy = a.idy;
    out[0] = a.aln, out[1] = a.mis, out[2] = a.gap, out[3] = a.qst, out[4] = a.qed, out[5] = a.sst, out[6] = a.sed,
    out[7] = a.bit, out[8] = a.cells, out[9] = a.maxscore;
}

i64 oc_score2bit(i64 s) { return score2bit(s); }
double oc_bit2e(i64 D, i64 li, i64 lj, i64 bit) { return bit2e(D, li, lj, bit); }
void oc_f2s(double e, char* out, int cap) { snprintf(out, (size_t)cap, "%s", f2s(e).c_str()); }
void oc_fmt_idy(double x, char* out, int cap) { snprintf(out, (size_t)cap, "%s", fmt_idy(x).c_str()); }

// ---- index object -----------------------------------------------------------------
struct OcIndex {
    Fasta fa;
    Params p;
    Index ix;
};

void* oc_index_build(const char* fasta_bytes, i64 nbytes, const char* ssd, const char* nr, i64 step, i64 NC, i64 start, i64 end) {
    OcIndex* o = new OcIndex();
    o->fa.from_bytes(fasta_bytes, nbytes);
    o->p.ssd = ssd, o->p.nr = nr, o->p.step = step, o->p.ht = NC;
    if (resolve_nc(o->p) < 1) {
        delete o;
        return nullptr;
    }
    o->ix.build(o->fa, o->p, start, end);
    return o;
}
void oc_index_free(void* h) { delete (OcIndex*)h; }
i64 oc_index_threshold(void* h) { return ((OcIndex*)h)->ix.threshold; }
void oc_index_set_threshold(void* h, i64 t) { ((OcIndex*)h)->ix.threshold = t; }
i64 oc_index_nlocus(void* h) { return (i64)((OcIndex*)h)->ix.locus.size(); }
i64 oc_index_nsoas(void* h) { return (i64)((OcIndex*)h)->ix.soas.size(); }
const uint32_t* oc_index_start(void* h) { return ((OcIndex*)h)->ix.start.data(); }
const uint32_t* oc_index_locus(void* h) { return ((OcIndex*)h)->ix.locus.data(); }
const uint32_t* oc_index_soas(void* h) { return ((OcIndex*)h)->ix.soas.data(); }

// find_msav_m on one (already masked) query.  Returns #candidates (4 x u32 each into
// cand, cap entries); tuples (3 x i32 per seed hit) up to tcap entries, *ntuples gets
// the true count; marks gets ql bytes.
i64 oc_find_msav_m(void* h, const char* q, i64 ql, uint32_t* cand, i64 cap, int32_t* tuples, i64 tcap, i64* ntuples, char* marks) {
    OcIndex* o = (OcIndex*)h;
    std::vector<Cand> out;
    std::vector<int32_t> tp;
    std::vector<char> mk;
    o->ix.find_msav_m(std::string(q, (size_t)ql), out, &tp, &mk);
    for (i64 i = 0; i < (i64)out.size() && i < cap; ++i)
        cand[4 * i] = out[i].subj, cand[4 * i + 1] = out[i].score, cand[4 * i + 2] = out[i].qi, cand[4 * i + 3] = out[i].qj;
    if (ntuples) *ntuples = (i64)tp.size() / 3;
    if (tuples)
        for (i64 i = 0; i < (i64)tp.size() && i < 3 * tcap; ++i) tuples[i] = tp[i];
    if (marks)
        for (i64 i = 0; i < (i64)mk.size() && i < ql; ++i) marks[i] = mk[i];
    return (i64)out.size();
}

// ---- end-to-end ----------------------------------------------------------------------
struct OcResult {
    std::vector<HitRec> recs;
    std::vector<std::vector<Cand>> cands;
    Stats st;
};

// params: expect, v, max_miss, st, ed, rst, red, thr, step, flt('T'/..), ht(NC), chk
void* oc_blastp(const char* qry, const char* ref, const char* ssd, const char* nr, double expect, i64 v, double max_miss, i64 st,
                i64 ed, i64 rst, i64 red, i64 thr, i64 step, const char* flt, i64 ht, i64 chk, const char* out_path,
                const char* mode) {
    Params p;
    p.ssd = ssd, p.nr = nr, p.expect = expect, p.v = v, p.max_miss = max_miss, p.st = st, p.ed = ed, p.rst = rst, p.red = red,
    p.thr = thr, p.step = step, p.flt = flt, p.ht = ht, p.chk = chk;
    OcResult* r = new OcResult();
    FILE* f = nullptr;
    if (out_path && out_path[0]) f = fopen(out_path, (mode && mode[0] == 'a') ? "ab" : "wb");
    int rc = blastp(qry, ref, p, f, r->st, &r->recs, &r->cands);
    if (f) fclose(f);
    if (rc) {
        delete r;
        return nullptr;
    }
    return r;
}
void oc_result_free(void* h) { delete (OcResult*)h; }
i64 oc_result_nrecs(void* h) { return (i64)((OcResult*)h)->recs.size(); }
// 15 fields per rec as doubles/ints: copy into parallel arrays
void oc_result_recs(void* h, i64* ints13, double* dbl2) {
    auto& v = ((OcResult*)h)->recs;
    for (size_t i = 0; i < v.size(); ++i) {
        const HitRec& r = v[i];
        i64* o = ints13 + 13 * i;
        o[0] = r.qidx, o[1] = r.sidx, o[2] = r.aln, o[3] = r.mis, o[4] = r.gap, o[5] = r.qst, o[6] = r.qed, o[7] = r.sst,
        o[8] = r.sed, o[9] = r.bit, o[10] = r.qlen, o[11] = r.slen, o[12] = r.ungapped;
        dbl2[2 * i] = r.idy, dbl2[2 * i + 1] = r.e;
    }
}
i64 oc_result_ncands(void* h, i64 qrel) { return (i64)((OcResult*)h)->cands[(size_t)qrel].size(); }
i64 oc_result_nqueries(void* h) { return (i64)((OcResult*)h)->cands.size(); }
void oc_result_cands(void* h, i64 qrel, uint32_t* out4) {
    auto& c = ((OcResult*)h)->cands[(size_t)qrel];
    for (size_t i = 0; i < c.size(); ++i) out4[4 * i] = c[i].subj, out4[4 * i + 1] = c[i].score, out4[4 * i + 2] = c[i].qi, out4[4 * i + 3] = c[i].qj;
}
// stats: n_queries, query_aa, rows, seed_hits, groups, ungap_steps, cands, alignments, cells ; times: index, seed, align
void oc_result_stats(void* h, i64* s9, double* t3) {
    Stats& s = ((OcResult*)h)->st;
    s9[0] = s.n_queries, s9[1] = s.query_aa, s9[2] = s.rows, s9[3] = s.seed_hits, s9[4] = s.groups, s9[5] = s.ungap_steps,
    s9[6] = s.cands, s9[7] = s.alignments, s9[8] = s.cells;
    t3[0] = s.t_index, t3[1] = s.t_seed, t3[2] = s.t_align;
}

}  // extern "C"

// =================================================================================
// `sohit_cpu` executable: same flags as the reference's lib/fsearch-c
// (entry_point, fsearch.py:3152-3264).  Built only with -DSOHIT_CPU_MAIN.
// =================================================================================
#ifdef SOHIT_CPU_MAIN
int main(int argc, char** argv) {
    std::unordered_map<std::string, std::string> args = {
        {"-p", ""},   {"-v", "500"}, {"-s", "111111"}, {"-i", ""},  {"-d", ""},  {"-e", "1e-3"}, {"-l", "-1"},
        {"-u", "-1"}, {"-m", "1e-3"}, {"-t", "-1"},    {"-r", "AST,CFILMVY,DN,EQ,G,H,KR,P,W"},   {"-j", "4"},
        {"-F", "T"},  {"-o", ""},    {"-D", ""},       {"-O", "wb"}, {"-L", "-1"}, {"-U", "-1"}, {"-M", "-1"},
        {"-c", "50000"}, {"-T", "./tmpdir"}};
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        if (args.count(k)) {
            if (i + 1 < argc) args[k] = argv[i + 1];
        } else if (k.size() > 2 && args.count(k.substr(0, 2))) {
            args[k.substr(0, 2)] = k.substr(2);
        }
    }
    if (args["-p"] != "blastp" || args["-i"].empty() || args["-d"].empty()) {
        printf("Usage:\n  sohit_cpu -p blastp -i qry.fsa -d db.fsa\n");
        return 0;
    }
    Params p;
    p.ssd = args["-s"], p.nr = args["-r"];
    p.expect = atof(args["-e"].c_str()), p.v = atoll(args["-v"].c_str()), p.st = atoll(args["-l"].c_str());
    p.ed = atoll(args["-u"].c_str()), p.rst = atoll(args["-L"].c_str()), p.red = atoll(args["-U"].c_str());
    p.max_miss = atof(args["-m"].c_str()), p.thr = atoll(args["-t"].c_str()), p.step = atoll(args["-j"].c_str());
    p.flt = args["-F"], p.ht = atoll(args["-M"].c_str()), p.chk = atoll(args["-c"].c_str());
    std::string wrt = args["-O"];
    FILE* f = args["-o"].empty() ? stdout : fopen(args["-o"].c_str(), (wrt.size() && wrt[0] == 'a') ? "ab" : "wb");
    if (!f) return 0;
    Stats st;
    blastp(args["-i"].c_str(), args["-d"].c_str(), p, f, st);
    if (f != stdout) fclose(f);
    fprintf(stderr,
            "sohit_cpu: queries=%lld aa=%lld rows=%lld seed_hits=%lld groups=%lld ungap_steps=%lld cands=%lld aln=%lld cells=%lld "
            "t_index=%.3f t_seed=%.3f t_align=%.3f\n",
            (long long)st.n_queries, (long long)st.query_aa, (long long)st.rows, (long long)st.seed_hits, (long long)st.groups,
            (long long)st.ungap_steps, (long long)st.cands, (long long)st.alignments, (long long)st.cells, st.t_index, st.t_seed,
            st.t_align);
    return 0;
}
#endif

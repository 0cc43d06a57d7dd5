// Native Kaldi minibatch loader (include/xvector_io.h): scp/ark index, 'CM ' sub-range decode, the reference's
// random speaker/segment sampler, a thread pool filling a ring of batches.  Host-only C++17 (no HIP).
//
// Sampling rules = dataset/data_loader.py:271-298 of the reference:
//   sample `num_speakers` distinct speakers -> ONE length T in [min_len, max_len] for the batch -> per speaker the
//   utterances with num_frames > T (none: replace the speaker by a random one outside the batch) -> `num_segments`
//   of them without replacement (list repeated when shorter) -> random start frame in [0, n - T].
// Unlike the reference (seeded from os.urandom) batch i is a pure function of (seed, i), so a run is reproducible
// for any number of threads; the consumer receives batches in index order.
#include "xvector_io.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local char g_err[1024] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

// ---------------------------------------------------------------------------------------------
// Kaldi matrix codec (kaldi_io.py:768-867 / compressed-matrix.h): all arithmetic in float, no contraction
// (-ffp-contract=off) so the result is bit-identical to the NumPy restatement and to the reference reader.
// ---------------------------------------------------------------------------------------------
const float kU16 = 1.52590218966964e-05f;   // 1/65535

struct ColHeader { uint16_t p0, p25, p75, p100; };

bool pread_all(int fd, void* buf, size_t n, off_t off) {
    char* p = (char*)buf;
    while (n > 0) {
        ssize_t r = pread(fd, p, n, off);
        if (r <= 0) return false;
        p += r; off += r; n -= (size_t)r;
    }
    return true;
}

// An ark file as the decoder reads it: a read-only shared mapping when the file can be mapped (the loader: dozens of decoder threads
// in up to eight processes read the SAME files; every pread takes and drops page-cache references, and eight loaders saturated at
// ~400 k chunks/s on the 256-CPU host whatever the thread count), pread otherwise.
struct Source {
    int fd = -1;
    const uint8_t* map = nullptr;
    size_t size = 0;
    bool get(void* dst, size_t n, int64_t off) const {
        if (!map) return pread_all(fd, dst, n, (off_t)off);
        if (off < 0 || (size_t)off + n > size) return false;
        memcpy(dst, map + off, n);
        return true;
    }
    // n bytes at off: a pointer into the mapping, or into `scratch` after a pread
    const uint8_t* view(size_t n, int64_t off, std::vector<uint8_t>& scratch) const {
        if (map) return (off >= 0 && (size_t)off + n <= size) ? map + off : nullptr;
        scratch.resize(n);
        return pread_all(fd, scratch.data(), n, (off_t)off) ? scratch.data() : nullptr;
    }
    // XVIO_NO_MMAP=1 keeps every read on pread: a mapped ark that is truncated or replaced while training runs (feature preparation
    // re-run, NFS) ends the process with SIGBUS in a decoder thread, pread returns a clean "truncated" error instead.  Arks must be
    // immutable while a loader that maps them is open (include/xvector_io.h).
    void open_map() {
        const char* no = getenv("XVIO_NO_MMAP");
        if (no && no[0] && no[0] != '0') return;
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0 || st.st_size <= 0) return;
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) return;
        map = (const uint8_t*)m; size = (size_t)st.st_size;
    }
    void close_all() {
        if (map) munmap((void*)map, size);
        if (fd >= 0) close(fd);
        map = nullptr; fd = -1; size = 0;
    }
};

// rows [start, start+length) of the matrix whose "\0B" marker sits at `off`; scratch is per-thread
int read_rows_fd(const Source& src, const char* name, int64_t off, int start, int length, float* out, int64_t capacity, int* rows_out,
                 int* cols_out, std::vector<uint8_t>& scratch) {
    char head[5];
    if (!src.get(head, 5, off)) return fail("%s:%lld: cannot read the matrix header", name, (long long)off);
    if (head[0] != '\0' || head[1] != 'B') return fail("%s:%lld: not a binary Kaldi object", name, (long long)off);
    off += 5;
    if (memcmp(head + 2, "CM ", 3) == 0) {
        struct { float minv, range; int32_t rows, cols; } g;
        if (!src.get(&g, 16, off)) return fail("%s: truncated CM header", name);
        off += 16;
        const int rows = g.rows, cols = g.cols;
        if (rows <= 0 || cols <= 0) return fail("%s: bad CM shape %d x %d", name, rows, cols);
        if (length < 0) length = rows - start;
        if (start < 0 || start + length > rows) return fail("The number of frames is not enough for length %d (%s: %d rows, start %d)", length, name, rows, start);
        if ((int64_t)length * cols > capacity) return fail("%s: output buffer too small", name);
        std::vector<ColHeader> ch(cols);
        if (!src.get(ch.data(), (size_t)cols * 8, off)) return fail("%s: truncated CM column headers", name);
        off += (int64_t)cols * 8;
        // column-major bytes: one read spanning the requested rows of every column
        const size_t span = (size_t)(cols - 1) * rows + length;
        const uint8_t* bytes = src.view(span, off + start, scratch);
        if (!bytes) return fail("%s: truncated CM data", name);
        const float gs = g.range * kU16;
        for (int c = 0; c < cols; ++c) {
            const float p0 = g.minv + gs * (float)ch[c].p0, p25 = g.minv + gs * (float)ch[c].p25;
            const float p75 = g.minv + gs * (float)ch[c].p75, p100 = g.minv + gs * (float)ch[c].p100;
            const float s_lo = (p25 - p0) / 64.0f, s_mid = (p75 - p25) / 128.0f, s_hi = (p100 - p75) / 63.0f;
            const uint8_t* col = bytes + (size_t)c * rows;
            for (int r = 0; r < length; ++r) {
                const uint8_t b = col[r];
                const float v = (float)b;
                float y;
                if (b <= 64) y = p0 + s_lo * v;
                else if (b <= 192) y = p25 + s_mid * (v - 64.0f);
                else y = p75 + s_hi * (v - 192.0f);
                out[(size_t)r * cols + c] = y;
            }
        }
        *rows_out = length; *cols_out = cols;
        return 0;
    }
    const bool fm = memcmp(head + 2, "FM ", 3) == 0, dm = memcmp(head + 2, "DM ", 3) == 0;
    if (!fm && !dm) return fail("%s: The header contained '%.3s'", name, head + 2);
    unsigned char dims[10];
    if (!src.get(dims, 10, off)) return fail("%s: truncated matrix header", name);
    off += 10;
    int32_t rows, cols;
    memcpy(&rows, dims + 1, 4); memcpy(&cols, dims + 6, 4);
    if (length < 0) length = rows - start;
    if (start < 0 || start + length > rows) return fail("The number of frames is not enough for length %d (%s: %d rows)", length, name, rows);
    if ((int64_t)length * cols > capacity) return fail("%s: output buffer too small", name);
    const size_t esz = fm ? 4 : 8;
    const size_t count = (size_t)length * cols;
    if (fm) {
        if (!src.get(out, count * 4, off + (int64_t)start * cols * 4)) return fail("%s: truncated FM data", name);
    } else {
        const uint8_t* bytes = src.view(count * 8, off + (int64_t)start * cols * 8, scratch);
        if (!bytes) return fail("%s: truncated DM data", name);
        for (size_t i = 0; i < count; ++i) { double d; memcpy(&d, bytes + 8 * i, 8); out[i] = (float)d; }
    }
    (void)esz;
    *rows_out = length; *cols_out = cols;
    return 0;
}

// Packed form of rows [start, start+length) of a 'CM ' matrix, for decoding on the GPU (xv_cm_decode, include/xvector_hip.h): the
// matrix header's (min, range), the per-column percentile headers and the rows' bytes, column after column - nothing is decoded here:
//   [min f32][range f32][cols x (p0, p25, p75, p100) u16][cols x length u8]      (xvio_packed_chunk_bytes: padded to 16 bytes)
int read_rows_packed(const Source& src, const char* name, int64_t off, int start, int length, uint8_t* out, int dim) {
    char head[5];
    if (!src.get(head, 5, off)) return fail("%s:%lld: cannot read the matrix header", name, (long long)off);
    if (head[0] != '\0' || head[1] != 'B') return fail("%s:%lld: not a binary Kaldi object", name, (long long)off);
    if (memcmp(head + 2, "CM ", 3) != 0) return fail("%s: the packed (GPU-decoded) loader needs 'CM ' compressed matrices, found '%.3s'", name, head + 2);
    off += 5;
    struct { float minv, range; int32_t rows, cols; } g;
    if (!src.get(&g, 16, off)) return fail("%s: truncated CM header", name);
    off += 16;
    if (g.cols != dim) return fail("feature dimension changes inside %s", name);
    if (start < 0 || length < 0 || start + length > g.rows) return fail("The number of frames is not enough for length %d (%s: %d rows, start %d)", length, name, g.rows, start);
    memcpy(out, &g.minv, 4); memcpy(out + 4, &g.range, 4);
    if (!src.get(out + 8, (size_t)dim * 8, off)) return fail("%s: truncated CM column headers", name);
    off += (int64_t)dim * 8;
    uint8_t* dst = out + 8 + (size_t)dim * 8;
    for (int c = 0; c < dim; ++c)
        if (!src.get(dst + (size_t)c * length, (size_t)length, off + (int64_t)c * g.rows + start)) return fail("%s: truncated CM data", name);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// random numbers: splitmix64-seeded xoshiro256**, one generator per batch index
// ---------------------------------------------------------------------------------------------
struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    Rng(uint64_t seed, uint64_t index) {
        uint64_t x = seed * 0xD1342543DE82EF95ull + index * 0x2545F4914F6CDD1Dull + 0x632BE59BD9B4E019ull;
        for (int i = 0; i < 4; ++i) s[i] = splitmix(x);
    }
    static uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    // uniform integer in [0, n)
    uint64_t below(uint64_t n) {
        const uint64_t lim = UINT64_MAX - UINT64_MAX % n;
        uint64_t v;
        do v = next(); while (v >= lim);
        return v % n;
    }
    int range(int lo, int hi) { return lo + (int)below((uint64_t)(hi - lo + 1)); }   // inclusive, like random.randint
};

struct Utt { int fd_index; int64_t offset; int num_frames; };

struct Slot {
    std::vector<float> features;
    std::vector<uint8_t> packed;      // packed mode: the chunks' undecoded 'CM ' pieces instead of `features`
    std::vector<int32_t> labels;
    int frames = 0;
    int64_t index = -1;     // batch index held (READY) or being filled
    int state = 0;          // 0 empty, 1 filling, 2 ready, 3 failed
    std::string error;
};

}  // namespace

struct xvio_loader {
    xvio_config cfg;
    std::string data_dir;
    int dim = 0;
    int total_speakers = 0;
    std::vector<Source> fds;       // one per ark file, mapped
    std::vector<std::string> ark_names;
    std::vector<Utt> utts;
    std::vector<std::vector<int>> spk_utts;   // per speaker (dense order of first appearance) -> utterance ids
    std::vector<int> spk_label;               // -> integer of spklist
    std::vector<int> speakers;                // sampling population (duplicated when fewer than num_speakers)

    std::vector<Slot> slots;
    std::mutex mu;
    std::condition_variable cv_ready, cv_free;
    std::atomic<int64_t> next_index{0};
    int64_t consume_index = 0;
    bool stopping = false;
    std::vector<std::thread> threads;
    std::atomic<int64_t> batches_done{0};
    std::atomic<int64_t> decode_ns{0};

    int fill(Slot& s, int64_t index, std::vector<uint8_t>& scratch);
    void worker();
};

int xvio_loader::fill(Slot& s, int64_t index, std::vector<uint8_t>& scratch) {
    Rng rng(cfg.seed, (uint64_t)index);
    const int S = cfg.num_speakers, G = cfg.num_segments;
    const int T = rng.range(cfg.min_len, cfg.max_len);
    // random.sample(speakers, S): partial Fisher-Yates over a copy of the population
    std::vector<int> pop(speakers);
    std::vector<int> batch(S);
    for (int i = 0; i < S; ++i) {
        int j = i + (int)rng.below(pop.size() - i);
        std::swap(pop[i], pop[j]);
        batch[i] = pop[i];
    }
    s.frames = T;
    std::vector<int> cand, picks;
    for (int i = 0; i < S; ++i) {
        int spk = batch[i];
        int guard = 0;
        for (;;) {
            cand.clear();
            for (int u : spk_utts[spk]) if (utts[u].num_frames > T) cand.push_back(u);
            if (!cand.empty()) break;
            // speakers outside the batch (set difference, as the reference does)
            std::vector<int> rest;
            for (size_t k = 0; k < spk_utts.size(); ++k)
                if (std::find(batch.begin(), batch.end(), (int)k) == batch.end()) rest.push_back((int)k);
            if (rest.empty() || ++guard > 10000) {
                s.error = "no speaker has an utterance longer than " + std::to_string(T) + " frames";
                return 1;
            }
            spk = rest[rng.below(rest.size())];
            batch[i] = spk;
        }
        // list repeated when shorter than num_segments, then a sample without replacement
        picks.clear();
        const int reps = (int)cand.size() < G ? G / (int)cand.size() + 1 : 1;
        for (int r = 0; r < reps; ++r) picks.insert(picks.end(), cand.begin(), cand.end());
        for (int j = 0; j < G; ++j) {
            int k = j + (int)rng.below(picks.size() - j);
            std::swap(picks[j], picks[k]);
            const Utt& u = utts[picks[j]];
            const int start = cfg.shuffle ? rng.range(0, u.num_frames - T) : 0;
            int rows = 0, cols = 0;
            if (cfg.packed) {
                uint8_t* dstp = s.packed.data() + (size_t)(i * G + j) * xvio_packed_chunk_bytes(dim, T);
                if (read_rows_packed(fds[u.fd_index], ark_names[u.fd_index].c_str(), u.offset, start, T, dstp, dim)) {
                    s.error = g_err;
                    return 1;
                }
                cols = dim;
            } else {
                float* dst = s.features.data() + (size_t)(i * G + j) * T * dim;
                if (read_rows_fd(fds[u.fd_index], ark_names[u.fd_index].c_str(), u.offset, start, T, dst, (int64_t)T * dim, &rows, &cols, scratch)) {
                    s.error = g_err;
                    return 1;
                }
            }
            if (cols != dim) { s.error = "feature dimension changes inside " + ark_names[u.fd_index]; return 1; }
            s.labels[i * G + j] = spk_label[spk];
        }
    }
    return 0;
}

void xvio_loader::worker() {
    std::vector<uint8_t> scratch;
    for (;;) {
        const int64_t index = next_index.fetch_add(1);
        Slot& s = slots[index % slots.size()];
        {
            std::unique_lock<std::mutex> lk(mu);
            // the slot is reusable once the consumer has taken batch index - depth
            cv_free.wait(lk, [&] { return stopping || (s.state == 0 && consume_index + (int64_t)slots.size() > index); });
            if (stopping) return;
            s.state = 1; s.index = index;
        }
        auto t0 = std::chrono::steady_clock::now();
        int rc = fill(s, index, scratch);
        decode_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        {
            std::lock_guard<std::mutex> lk(mu);
            s.state = rc ? 3 : 2;
        }
        batches_done++;
        cv_ready.notify_all();
    }
}

extern "C" const char* xvio_last_error(void) { return g_err; }
extern "C" int xvio_abi_version(void) { return 2; }

// CRC32C, slice-by-8 (tables built once, thread-safe static init)
extern "C" uint32_t xvio_crc32c(uint32_t crc, const void* data, uint64_t n) {
    struct Tables {
        uint32_t t[8][256];
        Tables() {
            for (uint32_t i = 0; i < 256; ++i) {
                uint32_t c = i;
                for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
                t[0][i] = c;
            }
            for (uint32_t i = 0; i < 256; ++i)
                for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
        }
    };
    static const Tables T;
    const uint8_t* p = (const uint8_t*)data;
    uint32_t c = ~crc;
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= c;
        c = T.t[7][w & 0xFF] ^ T.t[6][(w >> 8) & 0xFF] ^ T.t[5][(w >> 16) & 0xFF] ^ T.t[4][(w >> 24) & 0xFF] ^
            T.t[3][(w >> 32) & 0xFF] ^ T.t[2][(w >> 40) & 0xFF] ^ T.t[1][(w >> 48) & 0xFF] ^ T.t[0][(w >> 56) & 0xFF];
        p += 8; n -= 8;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}

extern "C" int xvio_read_rows(const char* ark_path, int64_t offset, int32_t start, int32_t length, float* out, int64_t capacity,
                              int32_t* rows_out, int32_t* cols_out) {
    if (!ark_path || !out || !rows_out || !cols_out) return fail("read_rows: null argument");
    int fd = open(ark_path, O_RDONLY);
    if (fd < 0) return fail("cannot open %s", ark_path);
    std::vector<uint8_t> scratch;
    int rows = 0, cols = 0;
    Source one;
    one.fd = fd;                  // a single sub-range read: plain pread, no mapping
    int rc = read_rows_fd(one, ark_path, offset, start, length, out, capacity, &rows, &cols, scratch);
    close(fd);
    *rows_out = rows; *cols_out = cols;
    return rc;
}

extern "C" int xvio_loader_create(const xvio_config* cfg, xvio_loader** out) {
    if (!cfg || !out || !cfg->data_dir || !cfg->spklist) return fail("loader_create: null argument");
    if (cfg->num_speakers < 1 || cfg->num_segments < 1 || cfg->min_len < 1 || cfg->max_len < cfg->min_len)
        return fail("loader_create: bad batch shape (speakers %d, segments %d, length [%d, %d])", cfg->num_speakers, cfg->num_segments,
                    cfg->min_len, cfg->max_len);
    std::unique_ptr<xvio_loader> l(new xvio_loader);
    l->cfg = *cfg;
    l->data_dir = cfg->data_dir;
    l->cfg.data_dir = nullptr; l->cfg.spklist = nullptr;
    const std::string dir = l->data_dir;

    std::unordered_map<std::string, int> spk2label;
    {
        std::ifstream f(cfg->spklist);
        if (!f) return fail("cannot open %s", cfg->spklist);
        std::string spk; int idx;
        while (f >> spk >> idx) spk2label[spk] = idx;
    }
    l->total_speakers = (int)spk2label.size();
    std::unordered_map<std::string, int> utt2dense;     // utterance -> dense speaker id
    std::unordered_map<std::string, int> dense_of;
    {
        std::ifstream f(dir + "/spk2utt");
        if (!f) return fail("cannot open %s/spk2utt", dir.c_str());
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream is(line);
            std::string spk, utt;
            if (!(is >> spk)) continue;
            auto it = spk2label.find(spk);
            if (it == spk2label.end()) return fail("speaker %s of spk2utt is not in %s", spk.c_str(), cfg->spklist);
            int dense;
            auto d = dense_of.find(spk);
            if (d == dense_of.end()) {
                dense = (int)l->spk_utts.size();
                dense_of[spk] = dense;
                l->spk_utts.emplace_back();
                l->spk_label.push_back(it->second);
            } else dense = d->second;
            while (is >> utt) utt2dense[utt] = dense;
        }
    }
    std::unordered_map<std::string, int> utt2frames;
    {
        std::ifstream f(dir + "/utt2num_frames");
        if (!f) return fail("[Error] Expect utt2num_frames exists in %s ", dir.c_str());
        std::string utt; int n;
        while (f >> utt >> n) utt2frames[utt] = n;
    }
    std::unordered_map<std::string, int> ark_index;
    {
        std::ifstream f(dir + "/feats.scp");
        if (!f) return fail("cannot open %s/feats.scp", dir.c_str());
        std::string line;
        while (std::getline(f, line)) {
            size_t sp = line.find(' ');
            if (sp == std::string::npos) continue;
            std::string utt = line.substr(0, sp), rx = line.substr(sp + 1);
            while (!rx.empty() && (rx.back() == '\n' || rx.back() == '\r' || rx.back() == ' ')) rx.pop_back();
            size_t colon = rx.rfind(':');
            if (colon == std::string::npos) return fail("feats.scp: '%s' is not path:offset", rx.c_str());
            std::string path = rx.substr(0, colon);
            int64_t off = atoll(rx.c_str() + colon + 1);
            auto sd = utt2dense.find(utt);
            if (sd == utt2dense.end()) return fail("utterance %s of feats.scp is not in spk2utt", utt.c_str());
            auto nf = utt2frames.find(utt);
            if (nf == utt2frames.end()) return fail("utterance %s of feats.scp is not in utt2num_frames", utt.c_str());
            int ai;
            auto a = ark_index.find(path);
            if (a == ark_index.end()) {
                int fd = open(path.c_str(), O_RDONLY);
                if (fd < 0) return fail("cannot open %s", path.c_str());
                ai = (int)l->fds.size();
                Source src;
                src.fd = fd;
                src.open_map();
                l->fds.push_back(src);
                l->ark_names.push_back(path);
                ark_index[path] = ai;
            } else ai = a->second;
            l->spk_utts[sd->second].push_back((int)l->utts.size());
            l->utts.push_back(Utt{ai, off, nf->second});
        }
    }
    if (l->utts.empty()) return fail("%s/feats.scp lists no utterance", dir.c_str());
    {   // feature dimension from the first matrix (one row)
        std::vector<uint8_t> scratch;
        std::vector<float> row(1 << 16);
        int rows = 0, cols = 0;
        const Utt& u = l->utts[0];
        if (read_rows_fd(l->fds[u.fd_index], l->ark_names[u.fd_index].c_str(), u.offset, 0, 1, row.data(), (int64_t)row.size(), &rows, &cols, scratch)) {
            for (Source& f : l->fds) f.close_all();
            return 1;
        }
        l->dim = cols;
    }
    // speakers with at least one utterance form the population (data_loader.py:240-243)
    for (size_t k = 0; k < l->spk_utts.size(); ++k)
        if (!l->spk_utts[k].empty()) l->speakers.push_back((int)k);
    if (l->speakers.empty()) return fail("no speaker has utterances");
    if ((int)l->speakers.size() < cfg->num_speakers) {
        fprintf(stderr, "[Warning] The number of available speakers are less than the required speaker. Some speakers will be duplicated.\n");
        std::vector<int> base(l->speakers);
        const int reps = cfg->num_speakers / (int)base.size() + 1;
        for (int r = 1; r < reps; ++r) l->speakers.insert(l->speakers.end(), base.begin(), base.end());
    }
    const int depth = std::max(1, cfg->queue_depth);
    const size_t B = (size_t)cfg->num_speakers * cfg->num_segments;
    l->slots.resize(depth);
    for (auto& s : l->slots) {
        if (cfg->packed) s.packed.resize(B * (size_t)xvio_packed_chunk_bytes(l->dim, cfg->max_len));
        else s.features.resize(B * (size_t)cfg->max_len * l->dim);
        s.labels.resize(B);
    }
    const int nt = std::max(1, cfg->num_threads);
    xvio_loader* raw = l.release();
    for (int i = 0; i < nt; ++i) raw->threads.emplace_back([raw] { raw->worker(); });
    *out = raw;
    return 0;
}

extern "C" void xvio_loader_destroy(xvio_loader* l) {
    if (!l) return;
    {
        std::lock_guard<std::mutex> lk(l->mu);
        l->stopping = true;
    }
    l->cv_free.notify_all();
    l->cv_ready.notify_all();
    for (auto& t : l->threads) t.join();
    for (Source& f : l->fds) f.close_all();
    delete l;
}

extern "C" int xvio_loader_dim(const xvio_loader* l) { return l ? l->dim : 0; }
extern "C" int xvio_loader_total_speakers(const xvio_loader* l) { return l ? l->total_speakers : 0; }
extern "C" int xvio_loader_num_utterances(const xvio_loader* l) { return l ? (int)l->utts.size() : 0; }

extern "C" int64_t xvio_packed_chunk_bytes(int32_t dim, int32_t frames) {
    return ((int64_t)8 + (int64_t)dim * 8 + (int64_t)dim * frames + 15) / 16 * 16;
}

static int loader_next_any(xvio_loader* l, void* features, int32_t* labels, int32_t* frames, bool packed);

extern "C" int xvio_loader_next(xvio_loader* l, float* features, int32_t* labels, int32_t* frames) {
    if (l && l->cfg.packed) return fail("loader_next: this loader was created in packed mode (use xvio_loader_next_packed)");
    return loader_next_any(l, features, labels, frames, false);
}

extern "C" int xvio_loader_next_packed(xvio_loader* l, uint8_t* packed, int32_t* labels, int32_t* frames) {
    if (l && !l->cfg.packed) return fail("loader_next_packed: this loader decodes on the host (xvio_config.packed = 0)");
    return loader_next_any(l, packed, labels, frames, true);
}

static int loader_next_any(xvio_loader* l, void* features, int32_t* labels, int32_t* frames, bool packed) {
    if (!l || !features || !labels || !frames) return fail("loader_next: null argument");
    Slot& s = l->slots[l->consume_index % l->slots.size()];
    {
        std::unique_lock<std::mutex> lk(l->mu);
        l->cv_ready.wait(lk, [&] { return l->stopping || ((s.state == 2 || s.state == 3) && s.index == l->consume_index); });
        if (l->stopping) return fail("loader_next: loader is shutting down");
    }
    int rc = 0;
    if (s.state == 3) rc = fail("%s", s.error.c_str());
    else {
        const size_t B = s.labels.size();
        if (packed) memcpy(features, s.packed.data(), B * (size_t)xvio_packed_chunk_bytes(l->dim, s.frames));
        else memcpy(features, s.features.data(), B * (size_t)s.frames * l->dim * sizeof(float));
        memcpy(labels, s.labels.data(), B * sizeof(int32_t));
        *frames = s.frames;
    }
    {
        std::lock_guard<std::mutex> lk(l->mu);
        s.state = 0;
        l->consume_index++;
    }
    l->cv_free.notify_all();
    return rc;
}

extern "C" int xvio_loader_stats(const xvio_loader* l, int64_t* batches, double* decode_seconds) {
    if (!l) return fail("loader_stats: null loader");
    if (batches) *batches = l->batches_done.load();
    if (decode_seconds) *decode_seconds = (double)l->decode_ns.load() * 1e-9;
    return 0;
}

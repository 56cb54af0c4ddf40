// fastx_ccs.hip -- stage 1 of `CIRI-long call` from file to file in native code (host side).
//
// What it replaces: the read loop of find_ccs_reads (CIRI_long/find_ccs.py:29-96): open FASTA/FASTQ(.gz) by suffix, one
// header line + one sequence line per record (FASTQ: two more lines skipped), header = first space-separated token
// without its leading '>' / '@' characters, consensus of every read, and the two tmp files in the reference's format
// (find_ccs.py:94-95):  tmp/{prefix}.ccs.fa  ">{header}\t{segments}\t{len(ccs)}\n{ccs}\n"   and
//                       tmp/{prefix}.raw.fa  ">{header}\n{raw sequence}\n"   -- reads with a consensus only, input order.
// The Python loop handles ~10^5 reads/s; K2+K3 handle ~4*10^6.  Here four threads work on six rotating batches of 32 MiB of file: a
// reader fills a batch (read(2); gzread for gzip files), a parser finds the records in place and encodes the bases, the calling thread
// keeps two batches on the device (plan, copies and launch of one under the kernels of the other), a writer formats and writes the two
// files.  The batches' buffers outlive the call (clh_ccs_file_release_buffers).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <algorithm>
#include <limits.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "../../include/ciri_long_hip.h"

namespace {

// a host buffer that outlives the call: a file stage of 10^5 reads does not spend a third of its time faulting in and unmapping
// 300 MB of fresh pages.  CLH_FILE_PINNED=1: page-locked (copies without staging: ~0.4 ms less per 32 MB batch, but ~1 ms per MB to
// allocate -- for a process that runs many files)
struct HostBuf {
    char* p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    bool ensure(size_t n)              // contents are not kept
    {
        if (n <= cap) return true;
        release();
        static const bool want_pinned = getenv("CLH_FILE_PINNED") != nullptr && atoi(getenv("CLH_FILE_PINNED")) != 0;
        const size_t c = n + (n >> 3) + 4096;
        if (want_pinned && hipHostMalloc((void**)&p, c, hipHostMallocDefault) == hipSuccess) pinned = true;
        else { if (want_pinned) (void)hipGetLastError(); p = (char*)malloc(c); pinned = false; }
        cap = p ? c : 0;
        return p != nullptr;
    }
    void release() { if (p) { if (pinned) (void)hipHostFree(p); else free(p); } p = nullptr; cap = 0; }
};

struct Batch {
    // the file's own bytes of this batch: `raw[lead .. lead + len)` holds whole records (the reader thread fills raw from kReserve on;
    // the parser puts the tail of the batch before -- a record cut by the chunk border -- in front of it).  Headers and sequences are
    // offsets into raw: nothing of the text is copied again
    HostBuf raw_buf;
    std::vector<char> raw_big;         // takes raw's place for one batch when a record is longer than the reserve (rare)
    char* raw = nullptr;
    size_t lead = 0, len = 0;
    bool eof = false;                  // the reader saw the end of the file while filling this batch
    std::vector<int64_t> hdr_off, hdr_len, seq_off, seq_len;   // into raw
    HostBuf codes_buf;                 // packed base codes of the reads handed to the GPU (never more than the batch has bytes)
    int8_t* codes = nullptr;
    size_t ncodes = 0;
    std::vector<int64_t> read_off;     // n_gpu + 1
    std::vector<int32_t> gpu_index;    // record -> row of the GPU batch, -1 = skipped (longer than the kernel's limit)
    bool last = false;
    void clear() { hdr_off.clear(); hdr_len.clear(); seq_off.clear(); seq_len.clear(); ncodes = 0; read_off.assign(1, 0); gpu_index.clear(); last = false; }
};
struct Results { HostBuf rows_buf, segs_buf, ccs_buf; int nrows = 0; };

// the four rotating batches, their results and the device side of two batches in flight; kept between calls (one set per process:
// a second call at the same time works on a set of its own)
struct FileCache {
    static const int NSLOT = 6;        // one being read, one parsed, two on the device, one written, one spare
    Batch slot[NSLOT];
    Results result[NSLOT];
    void* d_reads[2] = {nullptr, nullptr};
    size_t d_cap[2] = {0, 0};
    hipStream_t stream[2] = {nullptr, nullptr};
    int device = -1;
    void release()
    {
        for (auto& b : slot) { b.raw_buf.release(); b.codes_buf.release(); }
        for (auto& r : result) { r.rows_buf.release(); r.segs_buf.release(); r.ccs_buf.release(); }
        for (int i = 0; i < 2; ++i) { if (d_reads[i]) (void)hipFree(d_reads[i]); if (stream[i]) (void)hipStreamDestroy(stream[i]); d_reads[i] = nullptr; stream[i] = nullptr; d_cap[i] = 0; }
    }
};
std::mutex g_cache_mu;
FileCache* g_cache = nullptr;          // parked between calls

struct LineReader {
    gzFile f;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    explicit LineReader(gzFile f_) : f(f_), buf(1 << 22) {}
    // next line without its terminator; false at end of file
    bool next(std::string& line) {
        line.clear();
        for (;;) {
            if (pos == end) {
                if (eof) return !line.empty();
                const int got = gzread(f, buf.data(), (unsigned)buf.size());
                if (got <= 0) { eof = true; return !line.empty(); }
                pos = 0; end = (size_t)got;
            }
            const char* nl = (const char*)memchr(buf.data() + pos, '\n', end - pos);
            if (nl) { line.append(buf.data() + pos, nl - (buf.data() + pos)); pos = (size_t)(nl - buf.data()) + 1; return true; }
            line.append(buf.data() + pos, end - pos);
            pos = end;
        }
    }
    // the next line as a view into the buffer when it lies in it completely (nearly always: the buffer holds 4 MiB), otherwise
    // assembled in `tmp`; false at end of file.  The view is valid until the next call.
    bool next_view(const char*& p, size_t& n, std::string& tmp) {
        if (pos < end) {
            const char* nl = (const char*)memchr(buf.data() + pos, '\n', end - pos);
            if (nl) { p = buf.data() + pos; n = (size_t)(nl - p); pos = (size_t)(nl - buf.data()) + 1; return true; }
        }
        if (!next(tmp)) return false;
        p = tmp.data(); n = tmp.size();
        return true;
    }
    // read past one line; false at end of file
    bool skip_line() {
        bool any = false;
        for (;;) {
            if (pos == end) {
                if (eof) return any;
                const int got = gzread(f, buf.data(), (unsigned)buf.size());
                if (got <= 0) { eof = true; return any; }
                pos = 0; end = (size_t)got;
            }
            any = true;
            const char* nl = (const char*)memchr(buf.data() + pos, '\n', end - pos);
            if (nl) { pos = (size_t)(nl - buf.data()) + 1; return true; }
            pos = end;
        }
    }
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

const int kMaxRead = 1 << 24;          // sanity bound of clh_ccs_plan_create

// base codes -> letters of the consensus line ("ACGTN"[code], anything outside 0..4 reads N)
#if defined(__x86_64__)
__attribute__((target("avx2")))
void decode_avx2(char* dst, const int8_t* c, size_t n)
{
    size_t i = 0;
    const __m256i tab = _mm256_setr_epi8('A', 'C', 'G', 'T', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N');
    const __m256i four = _mm256_set1_epi8(4);
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i*)(c + i));
        const __m256i idx = _mm256_min_epu8(v, four);             // negative codes are large as unsigned: N
        _mm256_storeu_si256((__m256i*)(dst + i), _mm256_shuffle_epi8(tab, idx));
    }
    for (; i < n; ++i) dst[i] = "ACGTN"[c[i] < 0 || c[i] > 4 ? 4 : c[i]];
}
#endif
void decode_bases(char* dst, const int8_t* c, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { decode_avx2(dst, c, n); return; }
#endif
    for (size_t i = 0; i < n; ++i) dst[i] = "ACGTN"[c[i] < 0 || c[i] > 4 ? 4 : c[i]];
}
inline void put_int(std::string& s, int v)       // decimal digits of v (snprintf per number was a quarter of the writer thread's time)
{
    char t[16]; int k = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { t[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) t[k++] = '-';
    while (k) s.push_back(t[--k]);
}

// One read: encode the bases (ssw_wrap.py:50,243-250: A/a 0, C/c 1, G/g 2, T/t 3, anything else 4).  The table look-up per byte ran
// at 1 GB/s; as arithmetic on the letter's bits -- bits 1-2 of A C G T and of a c g t are 00 01 11 10 -- the loop vectorises
// (measured 6 GB/s with AVX2).  The text itself stays where the reader put it.
#if defined(__x86_64__)
__attribute__((target("avx2")))
void encode_avx2(int8_t* cd, const char* sp, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        const unsigned char ch = (unsigned char)sp[i], u = (unsigned char)(ch | 0x20);
        const unsigned char valid = (unsigned char)((u == 'a') | (u == 'c') | (u == 'g') | (u == 't'));
        const unsigned char x = (unsigned char)((ch >> 1) & 3);
        cd[i] = (int8_t)(valid ? (x ^ (x >> 1)) : 4);
    }
}
#endif
void encode_only(int8_t* cd, const char* sp, size_t n, const int8_t* lut)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { encode_avx2(cd, sp, n); return; }
#endif
    for (size_t i = 0; i < n; ++i) cd[i] = lut[(unsigned char)sp[i]];
}

}  // namespace

extern "C" int clh_fastx_count(const char* in_path, int is_fastq, int64_t* n_records)
{
    if (!in_path || !n_records) return CLH_E_ARG;
    gzFile in = gzopen(in_path, "rb");
    if (!in) return CLH_E_ARG;
    gzbuffer(in, 1 << 20);
    LineReader lr(in);
    std::string line;
    int64_t lines = 0;
    while (lr.skip_line()) ++lines;
    gzclose(in);
    const int per = is_fastq ? 4 : 2;
    *n_records = (lines + per - 1) / per;        // a trailing header without its sequence line is still a record (find_ccs.py:51-64)
    return 0;
}

// number of records, and the byte offset of every `every`-th record (records 0, every, 2 every, ...) of an UNCOMPRESSED file: a
// rank of a sharded run seeks to the indexed record at or before its shard instead of reading past everything in front of it.  A
// gzip file cannot be entered in the middle: *n_offsets = 0.  Host only.
extern "C" int clh_fastx_index(const char* in_path, int is_fastq, int64_t every, int64_t* n_records, int64_t* offsets, int64_t cap, int64_t* n_offsets)
{
    if (!in_path || !n_records || !n_offsets || every < 1 || (cap > 0 && !offsets)) return CLH_E_ARG;
    *n_offsets = 0;
    unsigned char magic[2] = {0, 0};
    FILE* probe = fopen(in_path, "rb");
    if (!probe) return CLH_E_ARG;
    const size_t got = fread(magic, 1, 2, probe);
    fclose(probe);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) return clh_fastx_count(in_path, is_fastq, n_records);
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) return CLH_E_ARG;
    std::vector<char> buf((size_t)8 << 20);
    const int per = is_fastq ? 4 : 2;
    int64_t lines = 0, pos = 0, no = 0;
    bool at_line_start = true, any_tail = false;
    for (;;) {
        const long n = (long)read(fd, buf.data(), buf.size());
        if (n <= 0) break;
        const char* p = buf.data();
        const char* const end = p + n;
        while (p < end) {
            if (at_line_start) {
                if (lines % per == 0 && (lines / per) % every == 0 && no < cap) offsets[no++] = pos + (int64_t)(p - buf.data());
                at_line_start = false; any_tail = true;
            }
            const char* e = (const char*)memchr(p, '\n', (size_t)(end - p));
            if (!e) break;
            ++lines; at_line_start = true; any_tail = false;
            p = e + 1;
        }
        pos += n;
    }
    close(fd);
    if (any_tail) ++lines;                       // a last line without its terminator
    *n_records = (lines + per - 1) / per;
    *n_offsets = no;
    return 0;
}

extern "C" int clh_ccs_file(clh_ctx* ctx, const char* in_path, int is_fastq, const char* ccs_fa_path, const char* raw_fa_path,
                            int32_t batch_reads, clh_ccs_file_stats* stats)
{
    return clh_ccs_file_at(ctx, in_path, is_fastq, ccs_fa_path, raw_fa_path, batch_reads, 0, 0, -1, stats);
}

extern "C" int clh_ccs_file_range(clh_ctx* ctx, const char* in_path, int is_fastq, const char* ccs_fa_path, const char* raw_fa_path,
                                  int32_t batch_reads, int64_t first_record, int64_t max_records, clh_ccs_file_stats* stats)
{
    return clh_ccs_file_at(ctx, in_path, is_fastq, ccs_fa_path, raw_fa_path, batch_reads, 0, first_record, max_records, stats);
}

extern "C" void clh_ccs_file_release_buffers(void)
{
    FileCache* c = nullptr;
    { std::lock_guard<std::mutex> g(g_cache_mu); c = g_cache; g_cache = nullptr; }
    if (c) { c->release(); delete c; }
}

// the records [first_record, first_record + max_records) counted from byte `byte_offset` of the file, which must be the first byte of a
// record (clh_fastx_index); a compressed file is read from its start whatever byte_offset says (the caller passes 0)
extern "C" int clh_ccs_file_at(clh_ctx* ctx, const char* in_path, int is_fastq, const char* ccs_fa_path, const char* raw_fa_path,
                               int32_t batch_reads, int64_t byte_offset, int64_t first_record, int64_t max_records, clh_ccs_file_stats* stats)
{
    if (!ctx || !in_path || !ccs_fa_path || !raw_fa_path || !stats || first_record < 0 || byte_offset < 0) return CLH_E_ARG;
    const double t_enter = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (batch_reads <= 0) batch_reads = 16384;      // small enough that the three stages overlap on files of 10^5 reads (measured: 65536 -> 0.54, 16384 -> 0.89 M reads/s)
    memset(stats, 0, sizeof(*stats));
    gzFile in = gzopen(in_path, "rb");
    if (!in) return CLH_E_ARG;
    gzbuffer(in, 1 << 20);
    FILE* fc = fopen(ccs_fa_path, "w");
    FILE* fr = fopen(raw_fa_path, "w");
    if (!fc || !fr) { if (fc) fclose(fc); if (fr) fclose(fr); gzclose(in); return CLH_E_ARG; }
    std::vector<char> wbuf1(1 << 22), wbuf2(1 << 22);
    setvbuf(fc, wbuf1.data(), _IOFBF, wbuf1.size());
    setvbuf(fr, wbuf2.data(), _IOFBF, wbuf2.size());

    int8_t lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = 4;                       // ssw_wrap.py:50,243-250: A/a C/c G/g T/t, anything else 4
    lut['A'] = lut['a'] = 0; lut['C'] = lut['c'] = 1; lut['G'] = lut['g'] = 2; lut['T'] = lut['t'] = 3;

    // six batches rotate through four threads: the READER fills a batch with file bytes (read(2) on a plain file, gzread on a
    // compressed one), the PARSER finds the records in them and encodes the bases, this thread runs the batch on the GPU, the WRITER
    // formats and writes the two files -- each takes the slots in order and waits for the state it consumes
    // (0 free -> 1 bytes read -> 2 parsed -> 3 computed -> 0).  Round 3 had one thread read, copy and encode: 1.3 GB/s of FASTQ.
    static const int NSLOT = FileCache::NSLOT;
    static const size_t kReserve = 4u << 20;         // room in front of a chunk for the record the previous chunk's border cut
    // the buffers: the set parked by the last call, or a new one (all page-locked allocations happen on this thread, which is the one
    // that has the context's device current)
    FileCache* fcache = nullptr;
    { std::lock_guard<std::mutex> g(g_cache_mu); fcache = g_cache; g_cache = nullptr; }
    if (!fcache) fcache = new FileCache();
    Batch* const slot = fcache->slot;
    Results* const result = fcache->result;
    auto park = [&]() {
        std::lock_guard<std::mutex> g(g_cache_mu);
        if (!g_cache) { g_cache = fcache; fcache = nullptr; }
        if (fcache) { fcache->release(); delete fcache; fcache = nullptr; }
    };
    std::mutex mu;
    std::condition_variable cv;
    int state[NSLOT] = {};
    std::atomic<bool> stop_reader{false};             // the parser has all it wants (max_records): the reader shall not wait for a slot again
    auto wait_state = [&](int s, int want) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return state[s] == want; }); };
    auto set_state = [&](int s, int v) { { std::lock_guard<std::mutex> lk(mu); state[s] = v; } cv.notify_all(); };
    // set under the mutex the reader's wait holds: a store between its predicate and its block would otherwise lose the wake-up
    auto tell_reader_to_stop = [&]() { { std::lock_guard<std::mutex> lk(mu); stop_reader = true; } cv.notify_all(); };
    // plain or gzip?  (the magic bytes; gzread serves both, but copies a plain file once more on the way)
    int fd = -1;
    {
        unsigned char magic[2] = {0, 0};
        FILE* probe = fopen(in_path, "rb");
        const size_t got = probe ? fread(magic, 1, 2, probe) : 0;
        if (probe) fclose(probe);
        if (!(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b)) fd = open(in_path, O_RDONLY);
        if (fd >= 0 && byte_offset > 0 && lseek(fd, (off_t)byte_offset, SEEK_SET) < 0) { park(); close(fd); fd = -1; fclose(fc); fclose(fr); gzclose(in); return CLH_E_ARG; }
        if (fd < 0 && byte_offset > 0) { park(); fclose(fc); fclose(fr); gzclose(in); return CLH_E_ARG; }      // a compressed file cannot be entered in the middle
    }
    size_t chunk_bytes = (size_t)32 << 20;
    if (const char* e = getenv("CLH_FILE_CHUNK_MB")) chunk_bytes = (size_t)std::max(1, atoi(e)) << 20;
    {
        // a plan of one read makes the context's device this thread's current one (and tells whether there is a GPU at all)
        const int64_t one[2] = {0, 1};
        clh_ccs_plan* probe = clh_ccs_plan_create(ctx, 1, one);
        if (!probe) { park(); if (fd >= 0) close(fd); fclose(fc); fclose(fr); gzclose(in); return CLH_E_HIP; }
        clh_ccs_plan_destroy(probe);
        int dev = -1;
        (void)hipGetDevice(&dev);
        if (fcache->device != dev) { fcache->release(); fcache->device = dev; }
        bool ok = true;
        for (int i = 0; i < NSLOT; ++i) ok = ok && slot[i].raw_buf.ensure(kReserve + chunk_bytes) && slot[i].codes_buf.ensure(kReserve + chunk_bytes + 64);
        for (int i = 0; i < 2 && ok; ++i) {
            const size_t need = kReserve + chunk_bytes + 256;
            if (fcache->d_cap[i] < need) {
                if (fcache->d_reads[i]) (void)hipFree(fcache->d_reads[i]);
                fcache->d_reads[i] = nullptr; fcache->d_cap[i] = 0;
                if (hipMalloc(&fcache->d_reads[i], need) == hipSuccess) fcache->d_cap[i] = need; else { (void)hipGetLastError(); ok = false; }
            }
            if (ok && !fcache->stream[i] && hipStreamCreateWithFlags(&fcache->stream[i], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ok = false; }
        }
        if (!ok) { park(); if (fd >= 0) close(fd); fclose(fc); fclose(fr); gzclose(in); return CLH_E_HIP; }
    }
    double t_read = 0, t_parse = 0, t_write = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto reader = [&]() {
        bool eof = false;
        for (int s = 0; !eof; s = (s + 1) % NSLOT) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return state[s] == 0 || stop_reader.load(); }); }
            if (stop_reader.load()) break;
            const double t0 = now();
            Batch& b = slot[s];
            b.raw = b.raw_buf.p;
            size_t have = 0;
            while (have < chunk_bytes) {
                const size_t want = chunk_bytes - have;
                long got;
                if (fd >= 0) got = (long)read(fd, b.raw + kReserve + have, want);
                else got = (long)gzread(in, b.raw + kReserve + have, (unsigned)std::min<size_t>(want, 1u << 30));
                if (got <= 0) { eof = true; break; }
                have += (size_t)got;
            }
            b.lead = kReserve; b.len = have; b.eof = eof;
            t_read += now() - t0;
            set_state(s, 1);
        }
    };
    auto parser = [&]() {
        std::vector<char> carry;                   // the bytes of a record the previous chunk's border cut
        const int per = is_fastq ? 4 : 2;
        int64_t to_skip = first_record;            // this rank's shard starts at record `first_record`: the records in front are read past
        int64_t left = max_records < 0 ? INT64_MAX : max_records;
        bool done = false;
        for (int s = 0; !done; s = (s + 1) % NSLOT) {
            wait_state(s, 1);
            const double t0 = now();
            Batch& b = slot[s];
            b.clear();
            b.codes = (int8_t*)b.codes_buf.p;
            if (!carry.empty()) {
                if (carry.size() > b.lead) {       // a record longer than the reserve: this batch in a buffer of its own, text then codes (rare)
                    const size_t text = carry.size() + b.len;
                    b.raw_big.assign(2 * text + 64, 0);
                    memcpy(b.raw_big.data() + carry.size(), b.raw + b.lead, b.len);
                    b.raw = b.raw_big.data();
                    b.lead = carry.size();
                    b.codes = (int8_t*)b.raw_big.data() + text;
                }
                b.lead -= carry.size();
                memcpy(b.raw + b.lead, carry.data(), carry.size());
                b.len += carry.size();
                carry.clear();
            }
            const char* base = b.raw;
            int8_t* const codes = b.codes;
            const char* p = base + b.lead;
            const char* const end = p + b.len;
            while (p < end && left > 0) {
                // the record's lines: whole lines only, unless the file ends here (then what is left is the last record, with the
                // lines it has: a header without its sequence line is a record with an empty sequence, find_ccs.py:51-64)
                const char* ls[4]; const char* le[4];
                int nl = 0;
                const char* q = p;
                while (nl < per && q < end) {
                    const char* e = (const char*)memchr(q, '\n', (size_t)(end - q));
                    if (!e) { if (!b.eof) break; ls[nl] = q; le[nl] = end; ++nl; q = end; break; }
                    ls[nl] = q; le[nl] = e; ++nl; q = e + 1;
                }
                if (nl < per && !b.eof) break;     // cut by the chunk border: the next chunk completes it
                if (nl == 0) break;
                p = q;
                if (to_skip > 0) { --to_skip; continue; }
                --left;
                const char* h0 = ls[0]; const char* h1 = le[0];
                while (h1 > h0 && is_space(h1[-1])) --h1;                           // str.rstrip
                const char* sp = (const char*)memchr(h0, ' ', (size_t)(h1 - h0));    // first space-separated token
                if (sp) h1 = sp;
                const char mark = is_fastq ? '@' : '>';
                while (h0 < h1 && *h0 == mark) ++h0;                                // str.lstrip: every leading marker character
                const char* s0 = nl > 1 ? ls[1] : end; const char* s1 = nl > 1 ? le[1] : end;
                while (s1 > s0 && is_space(s1[-1])) --s1;
                const size_t sn = (size_t)(s1 - s0);
                b.hdr_off.push_back((int64_t)(h0 - base)); b.hdr_len.push_back((int64_t)(h1 - h0));
                b.seq_off.push_back((int64_t)(s0 - base)); b.seq_len.push_back((int64_t)sn);
                if (sn == 0 || (int64_t)sn > kMaxRead) b.gpu_index.push_back(-1);
                else {
                    b.gpu_index.push_back((int32_t)(b.read_off.size() - 1));
                    encode_only(codes + b.ncodes, s0, sn, lut);
                    b.ncodes += sn;
                    b.read_off.push_back((int64_t)b.ncodes);
                }
            }
            if (left == 0 || (b.eof && p >= end)) done = true;
            else if (p < end) carry.assign(p, end);
            if (b.eof) done = true;
            b.last = done;
            t_parse += now() - t0;
            set_state(s, 2);
        }
        tell_reader_to_stop();         // nothing more will be parsed: the reader shall not fill further slots (up to six of 32 MB)
    };
    int wrc = 0;
    auto writer = [&]() {
        std::string oc, orw;                     // the two files' text of one batch: formatted in memory, written with one call each
        for (int s = 0;; s = (s + 1) % NSLOT) {
            wait_state(s, 3);
            const double tw0_ = now();
            Batch& b = slot[s];
            const Results& R = result[s];
            const clh_ccs_t* const rows = (const clh_ccs_t*)R.rows_buf.p;
            const int32_t* const segs = (const int32_t*)R.segs_buf.p;
            const int8_t* const ccs = (const int8_t*)R.ccs_buf.p;
            const int nrec = (int)b.hdr_off.size();
            oc.clear(); orw.clear();
            if (R.nrows > 0 || nrec > 0) {
                for (int k = 0; k < nrec; ++k) {
                    stats->total_reads += 1;
                    const int g = b.gpu_index[(size_t)k];
                    if (g < 0) { if (b.seq_len[(size_t)k] > kMaxRead) stats->too_long += 1; continue; }
                    if (R.nrows == 0) continue;                   // the GPU step failed: only the counters go on
                    const clh_ccs_t& r = rows[(size_t)g];
                    if (r.status != 0) { stats->capacity_dropped += 1; continue; }      // lost to a limit of the kernel: counted, reported by the caller
                    if (r.nseg <= 0) continue;
                    stats->ro_reads += 1;
                    const char* hdr = b.raw + b.hdr_off[(size_t)k];
                    const size_t hl = (size_t)b.hdr_len[(size_t)k];
                    oc.push_back('>'); oc.append(hdr, hl); oc.push_back('\t');
                    for (int i = 0; i < r.nseg; ++i) {
                        if (i) oc.push_back(';');
                        put_int(oc, segs[((size_t)g * 65 + (size_t)i) * 2]); oc.push_back('-'); put_int(oc, segs[((size_t)g * 65 + (size_t)i) * 2 + 1]);
                    }
                    oc.push_back('\t'); put_int(oc, r.ccs_len); oc.push_back('\n');
                    const size_t l0 = oc.size();
                    oc.resize(l0 + (size_t)r.ccs_len + 1);
                    decode_bases(&oc[l0], ccs + b.read_off[(size_t)g], (size_t)r.ccs_len);
                    oc[l0 + (size_t)r.ccs_len] = '\n';
                    orw.push_back('>'); orw.append(hdr, hl); orw.push_back('\n');
                    orw.append(b.raw + b.seq_off[(size_t)k], (size_t)b.seq_len[(size_t)k]); orw.push_back('\n');
                }
                if (!oc.empty() && fwrite(oc.data(), 1, oc.size(), fc) != oc.size()) wrc = CLH_E_ARG;
                if (!orw.empty() && fwrite(orw.data(), 1, orw.size(), fr) != orw.size()) wrc = CLH_E_ARG;
            }
            const bool last = b.last;
            t_write += now() - tw0_;
            set_state(s, 0);
            if (last) break;
        }
    };
    const bool ftrace = getenv("CLH_FILE_TRACE") != nullptr;
    double t_wait = 0, t_gpu = 0;
    const double t_start = now();
    std::thread th(reader), tp(parser), tw(writer);

    // This thread keeps TWO batches on the device: batch s is planned, copied and launched (its own stream) before the rows of batch
    // s - 1 are waited for and fetched, so planning and the copies of one batch run under the kernels of the other.
    int rc = 0;
    clh_ccs_plan* prev_pl = nullptr;
    void* prev_own = nullptr;
    int prev_slot = -1, lane = 0;
    auto complete = [&]() {           // the batch in flight: its rows to the writer
        if (prev_slot < 0) return;
        Results& R = result[prev_slot];
        if (prev_pl) {
            if (!rc) rc = clh_ccs_fetch(prev_pl, (clh_ccs_t*)R.rows_buf.p, (int32_t*)R.segs_buf.p, (int8_t*)R.ccs_buf.p);
            clh_ccs_plan_destroy(prev_pl);
            prev_pl = nullptr;
        }
        if (prev_own) { (void)hipFree(prev_own); prev_own = nullptr; }
        if (rc) R.nrows = 0;
        set_state(prev_slot, 3);
        prev_slot = -1;
    };
    for (int s = 0;; s = (s + 1) % NSLOT) {
        const double tw0 = now();
        wait_state(s, 2);
        const double tw1 = now();
        t_wait += tw1 - tw0;
        Batch& b = slot[s];
        Results& R = result[s];
        const int ngpu = (int)b.read_off.size() - 1;
        R.nrows = 0;
        clh_ccs_plan* pl = nullptr;
        void* own = nullptr;
        if (!rc && ngpu > 0) {
            if (!R.rows_buf.ensure(sizeof(clh_ccs_t) * (size_t)ngpu) || !R.segs_buf.ensure(sizeof(int32_t) * (size_t)ngpu * 2 * 65) || !R.ccs_buf.ensure(b.ncodes + 64)) rc = CLH_E_HIP;
            if (!rc && !(pl = clh_ccs_plan_create(ctx, ngpu, b.read_off.data()))) rc = CLH_E_ARG;
            void* d_reads = fcache->d_reads[lane];
            if (!rc && b.ncodes + 64 > fcache->d_cap[lane]) {       // (a batch in a buffer of its own, see the parser)
                if (hipMalloc(&own, b.ncodes + 64) != hipSuccess) { (void)hipGetLastError(); own = nullptr; rc = CLH_E_HIP; }
                d_reads = own;
            }
            if (!rc && b.ncodes > 0 && hipMemcpyAsync(d_reads, b.codes, b.ncodes, hipMemcpyHostToDevice, fcache->stream[lane]) != hipSuccess) { (void)hipGetLastError(); rc = CLH_E_HIP; }
            if (!rc) rc = clh_ccs_run(pl, d_reads, fcache->stream[lane]);
            if (!rc) R.nrows = ngpu;
        }
        const bool last = b.last;
        complete();                    // the batch before: its kernels ran while this one was planned, copied and launched
        prev_pl = pl; prev_own = own; prev_slot = s;
        lane ^= 1;
        if (last || own) complete();
        t_gpu += now() - tw1;
        if (last) break;
    }
    const double t_loop = now();
    tw.join();
    tp.join();
    // the parser may have stopped before the end of the file (max_records): a reader waiting for a free slot is told to stop
    tell_reader_to_stop();
    th.join();
    if (fd >= 0) close(fd);
    if (ftrace) fprintf(stderr, "[clh] file stage wall: set-up %.3f s, GPU thread's loop %.3f s, joins %.3f s\n", t_start - t_enter, t_loop - t_start, now() - t_loop);
    if (ftrace) fprintf(stderr, "[clh] file stage: read %.3f s, parse + encode %.3f s, GPU thread waited %.3f s, clh_ccs_batch %.3f s, format + write %.3f s\n", t_read, t_parse, t_wait, t_gpu, t_write);
    if (!rc && wrc) rc = wrc;
    for (int i = 0; i < NSLOT; ++i) { std::vector<char>().swap(slot[i].raw_big); }
    park();
    const double t_c0 = now();
    gzclose(in);
    if (fclose(fc) != 0 || fclose(fr) != 0) rc = rc ? rc : CLH_E_ARG;
    if (ftrace) fprintf(stderr, "[clh] file stage wall: closing the three files %.3f s; %.3f s since the call began\n", now() - t_c0, now() - t_enter);
    return rc;
}

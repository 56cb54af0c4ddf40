// fastx_ccs.hip -- stage 1 of `CIRI-long call` from file to file in native code (host side).
//
// What it replaces: the read loop of find_ccs_reads (CIRI_long/find_ccs.py:29-96): open FASTA/FASTQ(.gz) by suffix, one
// header line + one sequence line per record (FASTQ: two more lines skipped), header = first space-separated token
// without its leading '>' / '@' characters, consensus of every read, and the two tmp files in the reference's format
// (find_ccs.py:94-95):  tmp/{prefix}.ccs.fa  ">{header}\t{segments}\t{len(ccs)}\n{ccs}\n"   and
//                       tmp/{prefix}.raw.fa  ">{header}\n{raw sequence}\n"   -- reads with a consensus only, input order.
// The Python loop handles ~10^5 reads/s; K2+K3 handle ~4*10^6.  Here three threads work on three rotating batches: a reader
// parses and encodes (zlib's gzread serves plain and gzip files alike), the calling thread runs the batch on the GPU, a writer
// formats and writes the two files.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>
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

struct Batch {
    std::vector<char> text;            // headers and raw sequences, back to back
    std::vector<int64_t> hdr_off, hdr_len, seq_off, seq_len;   // into text
    std::vector<int8_t> codes;         // packed base codes of the reads handed to the GPU
    std::vector<int64_t> read_off;     // n_gpu + 1
    std::vector<int32_t> gpu_index;    // record -> row of the GPU batch, -1 = skipped (longer than the kernel's limit)
    bool last = false;
    void clear() { text.clear(); hdr_off.clear(); hdr_len.clear(); seq_off.clear(); seq_len.clear(); codes.clear(); read_off.assign(1, 0); gpu_index.clear(); last = false; }
};

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
inline void rstrip(std::string& s) { while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r' || s.back() == '\n' || s.back() == '\v' || s.back() == '\f')) s.pop_back(); }

const int kMaxRead = 1 << 24;          // sanity bound of clh_ccs_plan_create

// One read: keep the text, encode the bases (ssw_wrap.py:50,243-250: A/a 0, C/c 1, G/g 2, T/t 3, anything else 4).  The table
// look-up per byte ran at 1 GB/s and was two thirds of the reader thread's time; as arithmetic on the letter's bits -- bits 1-2 of
// A C G T and of a c g t are 00 01 11 10 -- the loop vectorises (measured 6 GB/s with AVX2).
#if defined(__x86_64__)
__attribute__((target("avx2")))
void encode_copy_avx2(char* td, int8_t* cd, const char* sp, size_t n)
{
    memcpy(td, sp, n);
    for (size_t i = 0; i < n; ++i) {
        const unsigned char ch = (unsigned char)sp[i], u = (unsigned char)(ch | 0x20);
        const unsigned char valid = (unsigned char)((u == 'a') | (u == 'c') | (u == 'g') | (u == 't'));
        const unsigned char x = (unsigned char)((ch >> 1) & 3);
        cd[i] = (int8_t)(valid ? (x ^ (x >> 1)) : 4);
    }
}
#endif
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

void encode_copy(char* td, int8_t* cd, const char* sp, size_t n, const int8_t* lut)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { encode_copy_avx2(td, cd, sp, n); return; }
#endif
    for (size_t i = 0; i < n; ++i) { const unsigned char ch = (unsigned char)sp[i]; td[i] = (char)ch; cd[i] = lut[ch]; }
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

extern "C" int clh_ccs_file(clh_ctx* ctx, const char* in_path, int is_fastq, const char* ccs_fa_path, const char* raw_fa_path,
                            int32_t batch_reads, clh_ccs_file_stats* stats)
{
    return clh_ccs_file_range(ctx, in_path, is_fastq, ccs_fa_path, raw_fa_path, batch_reads, 0, -1, stats);
}

extern "C" int clh_ccs_file_range(clh_ctx* ctx, const char* in_path, int is_fastq, const char* ccs_fa_path, const char* raw_fa_path,
                                  int32_t batch_reads, int64_t first_record, int64_t max_records, clh_ccs_file_stats* stats)
{
    if (!ctx || !in_path || !ccs_fa_path || !raw_fa_path || !stats || first_record < 0) return CLH_E_ARG;
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

    // three batches rotate through three threads: the reader parses and encodes, this thread runs the batch on the GPU, the
    // writer formats and writes the two files -- each takes the slots in order and waits for the state it consumes
    // (0 free -> 1 parsed -> 2 computed -> 0)
    static const int NSLOT = 3;
    struct Results { std::vector<clh_ccs_t> rows; std::vector<int32_t> segs; std::vector<int8_t> ccs; };
    Batch slot[NSLOT];
    Results result[NSLOT];
    std::mutex mu;
    std::condition_variable cv;
    int state[NSLOT] = {0, 0, 0};
    auto wait_state = [&](int s, int want) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return state[s] == want; }); };
    auto set_state = [&](int s, int v) { { std::lock_guard<std::mutex> lk(mu); state[s] = v; } cv.notify_all(); };
    auto reader = [&]() {
        LineReader lr(in);
        std::string header, seq;
        int s = 0;
        bool done = false;
        // this rank's shard starts at record `first_record`: the records in front are read past, not parsed
        for (int64_t k = 0; k < first_record && !done; ++k)
            for (int l = 0; l < (is_fastq ? 4 : 2); ++l) if (!lr.skip_line()) { done = true; break; }
        int64_t left = max_records < 0 ? INT64_MAX : max_records;
        if (left == 0) done = true;
        if (done) {       // nothing to do: hand the consumer an empty last batch
            wait_state(s, 0);
            slot[s].clear(); slot[s].last = true;
            set_state(s, 1);
        }
        while (!done) {
            wait_state(s, 0);
            Batch& b = slot[s];
            b.clear();
            while ((int)b.hdr_off.size() < batch_reads && b.codes.size() < (size_t)256 << 20) {
                if (left == 0) { done = true; break; }
                if (!lr.next(header)) { done = true; break; }
                --left;
                const char* sp_ = nullptr; size_t sn = 0;
                const bool have_seq = lr.next_view(sp_, sn, seq);       // NB: invalidates nothing of `header` (a std::string)
                (void)have_seq;     // a header line without a sequence line is a record with an empty sequence (as in the reference's loop)
                if (!have_seq) sn = 0;
                while (sn > 0 && is_space(sp_[sn - 1])) --sn;           // str.rstrip
                rstrip(header);
                size_t sp = header.find(' ');
                if (sp != std::string::npos) header.resize(sp);
                size_t lead = 0;
                const char mark = is_fastq ? '@' : '>';
                while (lead < header.size() && header[lead] == mark) ++lead;   // str.lstrip: every leading marker character
                b.hdr_off.push_back((int64_t)b.text.size()); b.hdr_len.push_back((int64_t)(header.size() - lead));
                b.text.insert(b.text.end(), header.begin() + (long)lead, header.end());
                b.seq_off.push_back((int64_t)b.text.size()); b.seq_len.push_back((int64_t)sn);
                const size_t t0 = b.text.size();
                b.text.resize(t0 + sn);
                if (sn == 0 || (int64_t)sn > kMaxRead) {
                    if (sn) memcpy(b.text.data() + t0, sp_, sn);
                    b.gpu_index.push_back(-1);
                } else {
                    b.gpu_index.push_back((int32_t)(b.read_off.size() - 1));
                    const size_t o = b.codes.size();
                    b.codes.resize(o + sn);
                    char* td = b.text.data() + t0; int8_t* cd = b.codes.data() + o;
                    encode_copy(td, cd, sp_, sn, lut);
                    b.read_off.push_back((int64_t)b.codes.size());
                }
                if (is_fastq) { lr.skip_line(); lr.skip_line(); }       // '+' line and qualities: read past, not copied
            }
            b.last = done;
            set_state(s, 1);
            s = (s + 1) % NSLOT;
        }
    };
    int wrc = 0;
    auto writer = [&]() {
        std::string oc, orw;                     // the two files' text of one batch: formatted in memory, written with one call each
        for (int s = 0;; s = (s + 1) % NSLOT) {
            wait_state(s, 2);
            Batch& b = slot[s];
            const Results& R = result[s];
            const int nrec = (int)b.hdr_off.size();
            oc.clear(); orw.clear();
            if (!R.rows.empty() || nrec > 0) {
                for (int k = 0; k < nrec; ++k) {
                    stats->total_reads += 1;
                    const int g = b.gpu_index[(size_t)k];
                    if (g < 0) { if (b.seq_len[(size_t)k] > kMaxRead) stats->too_long += 1; continue; }
                    if (R.rows.empty()) continue;                 // the GPU step failed: only the counters go on
                    const clh_ccs_t& r = R.rows[(size_t)g];
                    if (r.status != 0) { stats->capacity_dropped += 1; continue; }      // lost to a limit of the kernel: counted, reported by the caller
                    if (r.nseg <= 0) continue;
                    stats->ro_reads += 1;
                    const char* hdr = b.text.data() + b.hdr_off[(size_t)k];
                    const size_t hl = (size_t)b.hdr_len[(size_t)k];
                    oc.push_back('>'); oc.append(hdr, hl); oc.push_back('\t');
                    for (int i = 0; i < r.nseg; ++i) {
                        if (i) oc.push_back(';');
                        put_int(oc, R.segs[((size_t)g * 65 + (size_t)i) * 2]); oc.push_back('-'); put_int(oc, R.segs[((size_t)g * 65 + (size_t)i) * 2 + 1]);
                    }
                    oc.push_back('\t'); put_int(oc, r.ccs_len); oc.push_back('\n');
                    const size_t l0 = oc.size();
                    oc.resize(l0 + (size_t)r.ccs_len + 1);
                    decode_bases(&oc[l0], R.ccs.data() + b.read_off[(size_t)g], (size_t)r.ccs_len);
                    oc[l0 + (size_t)r.ccs_len] = '\n';
                    orw.push_back('>'); orw.append(hdr, hl); orw.push_back('\n');
                    orw.append(b.text.data() + b.seq_off[(size_t)k], (size_t)b.seq_len[(size_t)k]); orw.push_back('\n');
                }
                if (!oc.empty() && fwrite(oc.data(), 1, oc.size(), fc) != oc.size()) wrc = CLH_E_ARG;
                if (!orw.empty() && fwrite(orw.data(), 1, orw.size(), fr) != orw.size()) wrc = CLH_E_ARG;
            }
            const bool last = b.last;
            set_state(s, 0);
            if (last) break;
        }
    };
    const bool ftrace = getenv("CLH_FILE_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_wait = 0, t_gpu = 0;
    std::thread th(reader), tw(writer);

    int rc = 0;
    for (int s = 0;; s = (s + 1) % NSLOT) {
        const double tw0 = now();
        wait_state(s, 1);
        const double tw1 = now();
        t_wait += tw1 - tw0;
        Batch& b = slot[s];
        Results& R = result[s];
        const int ngpu = (int)b.read_off.size() - 1;
        R.rows.clear();
        if (!rc && ngpu > 0) {
            R.rows.resize((size_t)ngpu); R.segs.resize((size_t)ngpu * 2 * 65); R.ccs.resize(b.codes.size() + 64);
            rc = clh_ccs_batch(ctx, ngpu, b.codes.data(), b.read_off.data(), R.rows.data(), R.segs.data(), R.ccs.data());
            if (rc) R.rows.clear();
        }
        t_gpu += now() - tw1;
        const bool last = b.last;
        set_state(s, 2);
        if (last) break;
    }
    tw.join();
    th.join();
    if (ftrace) fprintf(stderr, "[clh] file stage: GPU thread waited for the reader %.3f s, clh_ccs_batch %.3f s\n", t_wait, t_gpu);
    if (!rc && wrc) rc = wrc;
    gzclose(in);
    if (fclose(fc) != 0 || fclose(fr) != 0) rc = rc ? rc : CLH_E_ARG;
    return rc;
}

// csc_json.cpp -- host side of libcsc_hip.so: streaming writer of the sparVar result list.
//
// The reference ends its pipeline with json.dump(results, f, indent=4, default=handle_non_serializable)
// (src/csc/dictionary.py:1148-1150, src/csc/utils.py:7-15): at 10^5..10^6 signals that call, not the solve, is the
// time.  csc_sparvar_json_write emits the SAME BYTES for the list of per-signal entries straight from the packed
// device records (one contiguous record per signal, CSC_FLAG_RECORDS_AOS), formatting floats like Python's
// float.__repr__ (shortest round-trip digits; fixed notation for 1e-4 <= |x| < 1e16, else d.ddde-XX) and reusing
// one pre-formatted block per dictionary atom.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include "../../include/csc_hip.h"

namespace {

// Python's repr(float): shortest digits that round-trip, laid out by float_repr_style 'short' (format code 'r'):
// exponent form iff decpt <= -4 or decpt > 16, where value = 0.DIGITS x 10^decpt.
inline char* py_float_repr(char* out, double x) {
    if (std::isnan(x)) { memcpy(out, "NaN", 3); return out + 3; }          // json.dump(allow_nan=True)
    if (std::isinf(x)) {
        if (x < 0) { memcpy(out, "-Infinity", 9); return out + 9; }
        memcpy(out, "Infinity", 8); return out + 8;
    }
    if (x == 0.0) {
        if (std::signbit(x)) *out++ = '-';
        memcpy(out, "0.0", 3); return out + 3;
    }
    char tmp[40];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, x, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
    const char* p = tmp;
    if (*p == '-') { *out++ = '-'; ++p; }
    char digits[24];
    int nd = 0;
    digits[nd++] = *p++;
    if (*p == '.') { ++p; while (*p != 'e') digits[nd++] = *p++; }
    ++p;                                                                   // 'e'
    int e10 = 0;
    const bool neg = *p == '-';
    ++p;                                                                   // sign
    while (p < r.ptr) e10 = e10 * 10 + (*p++ - '0');
    if (neg) e10 = -e10;
    const int decpt = e10 + 1;
    if (decpt <= -4 || decpt > 16) {                                       // exponent form
        *out++ = digits[0];
        if (nd > 1) { *out++ = '.'; memcpy(out, digits + 1, nd - 1); out += nd - 1; }
        *out++ = 'e';
        int e = decpt - 1;
        *out++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        if (e >= 100) { *out++ = (char)('0' + e / 100); e %= 100; *out++ = (char)('0' + e / 10); *out++ = (char)('0' + e % 10); }
        else { *out++ = (char)('0' + e / 10); *out++ = (char)('0' + e % 10); }
        return out;
    }
    if (decpt <= 0) {                                                      // 0.000ddd
        *out++ = '0'; *out++ = '.';
        for (int i = 0; i < -decpt; ++i) *out++ = '0';
        memcpy(out, digits, nd); return out + nd;
    }
    if (decpt >= nd) {                                                     // ddd000.0
        memcpy(out, digits, nd); out += nd;
        for (int i = nd; i < decpt; ++i) *out++ = '0';
        *out++ = '.'; *out++ = '0';
        return out;
    }
    memcpy(out, digits, decpt); out += decpt;                              // dd.ddd
    *out++ = '.';
    memcpy(out, digits + decpt, nd - decpt);
    return out + (nd - decpt);
}

inline char* put_int(char* out, long long v) {
    auto r = std::to_chars(out, out + 24, v);
    return r.ptr;
}

// everything of a record that does not depend on the record: indentation, keys, one tail per dictionary atom
struct Formatter {
    int K, connections, n_atoms;
    const char* records;
    const int64_t* ids;
    const double* snr;
    const uint8_t* snr_is_int;
    size_t rec_bytes, max_bytes;
    std::vector<std::string> atom_tail;
    std::string open_sig, snr_key, results_key, open_res, path_key, atoms_key, open_atom, delay_line, close_sig;

    // record s at p (followed by ",\n" unless it is the last of the list); nullptr = an atom index outside the dictionary
    char* record(char* p, int64_t s, bool last) const {
        const char* rec = records + (size_t)s * rec_bytes;
        const double* mse = reinterpret_cast<const double*>(rec);
        const int32_t* leaf = reinterpret_cast<const int32_t*>(rec + 8 * (size_t)K);
        const int32_t* sel = reinterpret_cast<const int32_t*>(rec + 16 * (size_t)K);
        memcpy(p, open_sig.data(), open_sig.size()); p += open_sig.size();
        p = put_int(p, ids[s]);
        memcpy(p, snr_key.data(), snr_key.size()); p += snr_key.size();
        if (snr_is_int && snr_is_int[s]) p = put_int(p, (long long)snr[s]);
        else p = py_float_repr(p, snr[s]);
        memcpy(p, results_key.data(), results_key.size()); p += results_key.size();
        for (int k = 1; k <= K; ++k) {
            memcpy(p, open_res.data(), open_res.size()); p += open_res.size();
            p = py_float_repr(p, mse[k - 1]);
            memcpy(p, path_key.data(), path_key.size()); p += path_key.size();
            int j = leaf[2 * (k - 1)];                                // the '<=' (last minimum) leaf
            for (int e = 0; e < k; ++e) {                             // base-c digits, least significant first, + 1
                if (e) *p++ = '-';
                p = put_int(p, j % connections + 1);
                j /= connections;
            }
            memcpy(p, atoms_key.data(), atoms_key.size()); p += atoms_key.size();
            for (int e = 0; e < k; ++e) {
                const int32_t pos = sel[((size_t)(k - 1) * K + e) * 2], atom = sel[((size_t)(k - 1) * K + e) * 2 + 1];
                if (atom < 0 || atom >= n_atoms) return nullptr;
                if (e) { *p++ = ','; *p++ = '\n'; }
                memcpy(p, open_atom.data(), open_atom.size()); p += open_atom.size();
                p = put_int(p, pos);
                const std::string& t = atom_tail[atom];
                memcpy(p, t.data(), t.size()); p += t.size();
            }
            memcpy(p, delay_line.data(), delay_line.size()); p += delay_line.size();
            if (k < K) { *p++ = ','; *p++ = '\n'; }
        }
        memcpy(p, close_sig.data(), close_sig.size()); p += close_sig.size();
        if (!last) { *p++ = ','; *p++ = '\n'; }
        return p;
    }
};

bool write_all(int fd, const char* buf, size_t n, off_t off) {
    while (n) {
        const ssize_t w = pwrite(fd, buf, n, off);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        buf += w; n -= (size_t)w; off += w;
    }
    return true;
}

thread_local char g_json_err[256] = "";

}  // namespace

extern "C" {

const char* csc_json_last_error(void) { return g_json_err; }

int csc_format_double(double x, char* out, int cap) {
    char tmp[48];
    char* e = py_float_repr(tmp, x);
    const int n = (int)(e - tmp);
    if (!out || cap <= n) return -1;
    memcpy(out, tmp, n);
    out[n] = 0;
    return n;
}

int csc_sparvar_json_write(const char* path, int append, const void* records, int64_t S, int max_sparsity,
                           int connections, const int64_t* ids, const double* snr, const uint8_t* snr_is_int,
                           double delay, const double* atom_b, const double* atom_y, const double* atom_s,
                           int n_atoms, int indent_level, int close_object) {
#define JFAIL(code, ...) do { snprintf(g_json_err, sizeof g_json_err, __VA_ARGS__); return (code); } while (0)
    const int K = max_sparsity;
    if (!path || (S > 0 && (!records || !ids || !snr)) || K < 1 || K > CSC_MAX_SPARSITY || connections < 1 ||
        !atom_b || !atom_y || !atom_s || n_atoms < 1 || indent_level < 0 || indent_level > 8)
        JFAIL(-1, "csc_sparvar_json_write: bad arguments");
    const int fd = open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0644);
    if (fd < 0) JFAIL(-2, "csc_sparvar_json_write: cannot open %s", path);
    off_t off = append ? lseek(fd, 0, SEEK_END) : 0;
    if (off < 0) { close(fd); JFAIL(-2, "csc_sparvar_json_write: cannot seek in %s", path); }
    // indentation strings: the list sits at `indent_level` (its '[' follows the key on the caller's line)
    auto ind = [&](int extra) { return std::string(4 * (size_t)(indent_level + extra), ' '); };
    const std::string i1 = ind(1), i2 = ind(2), i3 = ind(3), i4 = ind(4), i5 = ind(5), i6 = ind(6), i0 = ind(0);
    Formatter F;
    F.K = K; F.connections = connections; F.n_atoms = n_atoms;
    F.records = static_cast<const char*>(records); F.ids = ids; F.snr = snr; F.snr_is_int = snr_is_int;
    F.rec_bytes = (size_t)K * (16 + 8 * (size_t)K);
    // one pre-formatted tail per atom: ,\n"b": ..,\n"y": ..,\n"s": ..\n}   (everything after the "x" line)
    F.atom_tail.resize((size_t)n_atoms);
    char num[48];
    for (int k = 0; k < n_atoms; ++k) {
        std::string& t = F.atom_tail[k];
        t = ",\n" + i6 + "\"b\": ";
        t.append(num, py_float_repr(num, atom_b[k]) - num);
        t += ",\n" + i6 + "\"y\": ";
        t.append(num, py_float_repr(num, atom_y[k]) - num);
        t += ",\n" + i6 + "\"s\": ";
        t.append(num, py_float_repr(num, atom_s[k]) - num);
        t += "\n" + i5 + "}";
    }
    F.delay_line = "\n" + i4 + "],\n" + i4 + "\"delay\": ";
    F.delay_line.append(num, py_float_repr(num, delay) - num);
    F.delay_line += "\n" + i3 + "}";
    F.open_sig = i1 + "{\n" + i2 + "\"id\": ";
    F.snr_key = ",\n" + i2 + "\"snr\": ";
    F.results_key = ",\n" + i2 + "\"results\": [\n";
    F.open_res = i3 + "{\n" + i4 + "\"mse\": ";
    F.path_key = ",\n" + i4 + "\"path\": \"";
    F.atoms_key = "\",\n" + i4 + "\"atoms\": [\n";
    F.open_atom = i5 + "{\n" + i6 + "\"x\": ";
    F.close_sig = "\n" + i2 + "]\n" + i1 + "}";
    // worst case of one signal entry, from the pieces themselves (an integer takes <= 21 characters, a float <= 25)
    size_t tail_max = 0;
    for (const std::string& t : F.atom_tail) tail_max = std::max(tail_max, t.size());
    const size_t atom_max = 2 + F.open_atom.size() + 21 + tail_max;
    const size_t result_max = F.open_res.size() + 25 + F.path_key.size() + 22 * (size_t)K + F.atoms_key.size() +
                              (size_t)K * atom_max + F.delay_line.size() + 2;
    F.max_bytes = F.open_sig.size() + 21 + F.snr_key.size() + 25 + F.results_key.size() + (size_t)K * result_max +
                  F.close_sig.size() + 2 + 64;

    bool ok = true;
    int64_t bad_record = -1;
    auto put = [&](const std::string& t) { ok = ok && write_all(fd, t.data(), t.size(), off); off += (off_t)t.size(); };
    // formatting runs at ~1 M records/s per core and the page cache takes ~2 GB/s per writer, so large lists are cut
    // into chunks that threads format into buffers of their own and write at their own offsets (pwrite)
    int T = 1;
    if (S >= 8192) {
        T = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
        if (const char* e = getenv("CSC_JSON_THREADS")) T = std::max(1, std::min(64, atoi(e)));
    }
    if (S == 0) {
        put("[]");
    } else {
        put("[\n");
        const int64_t chunk = T > 1 ? 1024 : 512;                    // records per buffer
        std::vector<std::vector<char>> buf((size_t)T);
        for (auto& b : buf) b.resize((size_t)chunk * F.max_bytes);
        std::vector<size_t> len((size_t)T);
        std::atomic<int64_t> bad{-1};
        for (int64_t r0 = 0; r0 < S && ok && bad.load() < 0; r0 += chunk * T) {
            auto format = [&](int t) {
                const int64_t a = std::min(S, r0 + (int64_t)t * chunk), b = std::min(S, a + chunk);
                char* p = buf[t].data();
                for (int64_t s = a; s < b; ++s) {
                    char* q = F.record(p, s, s + 1 == S);
                    if (!q) { bad.store(s); break; }
                    p = q;
                }
                len[t] = (size_t)(p - buf[t].data());
            };
            std::vector<off_t> at((size_t)T);
            std::atomic<bool> wok{true};
            auto flush = [&](int t) { if (len[t] && !write_all(fd, buf[t].data(), len[t], at[t])) wok.store(false); };
            if (T == 1) {
                format(0);
                at[0] = off;
                if (bad.load() < 0) flush(0);
                off += (off_t)len[0];
            } else {
                std::vector<std::thread> th;
                for (int t = 0; t < T; ++t) th.emplace_back(format, t);
                for (auto& x : th) x.join();
                if (bad.load() >= 0) break;
                for (int t = 0; t < T; ++t) { at[t] = off; off += (off_t)len[t]; }
                th.clear();
                for (int t = 0; t < T; ++t) th.emplace_back(flush, t);
                for (auto& x : th) x.join();
            }
            ok = ok && wok.load();
        }
        bad_record = bad.load();
        if (bad_record < 0) put("\n" + i0 + "]");
    }
    if (bad_record < 0 && close_object) {                             // the enclosing dict(s) of the caller's header
        for (int lv = indent_level - 1; lv >= 0; --lv) put("\n" + std::string(4 * (size_t)lv, ' ') + "}");
    }
    ok = (close(fd) == 0) && ok;
    if (bad_record >= 0) JFAIL(-3, "record %lld: atom index outside the dictionary", (long long)bad_record);
    if (!ok) JFAIL(-4, "csc_sparvar_json_write: write to %s failed", path);
    return 0;
#undef JFAIL
}

}  // extern "C"

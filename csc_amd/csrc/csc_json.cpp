// csc_json.cpp -- host side of libcsc_hip.so: streaming writer of the sparVar result list.
//
// The reference ends its pipeline with json.dump(results, f, indent=4, default=handle_non_serializable)
// (src/csc/dictionary.py:1148-1150, src/csc/utils.py:7-15): at 10^5..10^6 signals that call, not the solve, is the
// time.  csc_sparvar_json_write emits the SAME BYTES for the list of per-signal entries straight from the packed
// device records (one contiguous record per signal, CSC_FLAG_RECORDS_AOS), formatting floats like Python's
// float.__repr__ (shortest round-trip digits; fixed notation for 1e-4 <= |x| < 1e16, else d.ddde-XX) and reusing
// one pre-formatted block per dictionary atom.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

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

struct Sink {
    FILE* f;
    std::vector<char> buf;
    size_t n = 0;
    bool ok = true;
    explicit Sink(FILE* file) : f(file), buf(4u << 20) {}
    char* room(size_t need) {
        if (n + need > buf.size()) flush();
        return buf.data() + n;
    }
    void advance(char* end) { n = (size_t)(end - buf.data()); }
    void put(const char* s, size_t len) { char* p = room(len); memcpy(p, s, len); n += len; }
    void put(const std::string& s) { put(s.data(), s.size()); }
    void flush() {
        if (n && fwrite(buf.data(), 1, n, f) != n) ok = false;
        n = 0;
    }
};

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
    FILE* f = fopen(path, append ? "ab" : "wb");
    if (!f) JFAIL(-2, "csc_sparvar_json_write: cannot open %s", path);
    Sink out(f);
    // indentation strings: the list sits at `indent_level` (its '[' follows the key on the caller's line)
    auto ind = [&](int extra) { return std::string(4 * (size_t)(indent_level + extra), ' '); };
    const std::string i1 = ind(1), i2 = ind(2), i3 = ind(3), i4 = ind(4), i5 = ind(5), i6 = ind(6), i0 = ind(0);
    // one pre-formatted tail per atom: ,\n"b": ..,\n"y": ..,\n"s": ..\n}   (everything after the "x" line)
    std::vector<std::string> atom_tail((size_t)n_atoms);
    char num[48];
    for (int k = 0; k < n_atoms; ++k) {
        std::string& t = atom_tail[k];
        t = ",\n" + i6 + "\"b\": ";
        t.append(num, py_float_repr(num, atom_b[k]) - num);
        t += ",\n" + i6 + "\"y\": ";
        t.append(num, py_float_repr(num, atom_y[k]) - num);
        t += ",\n" + i6 + "\"s\": ";
        t.append(num, py_float_repr(num, atom_s[k]) - num);
        t += "\n" + i5 + "}";
    }
    std::string delay_line = "\n" + i4 + "],\n" + i4 + "\"delay\": ";
    delay_line.append(num, py_float_repr(num, delay) - num);
    delay_line += "\n" + i3 + "}";
    const std::string open_sig = i1 + "{\n" + i2 + "\"id\": ";
    const std::string snr_key = ",\n" + i2 + "\"snr\": ";
    const std::string results_key = ",\n" + i2 + "\"results\": [\n";
    const std::string open_res = i3 + "{\n" + i4 + "\"mse\": ";
    const std::string path_key = ",\n" + i4 + "\"path\": \"";
    const std::string atoms_key = "\",\n" + i4 + "\"atoms\": [\n";
    const std::string open_atom = i5 + "{\n" + i6 + "\"x\": ";
    const std::string close_sig = "\n" + i2 + "]\n" + i1 + "}";
    const size_t rec_bytes = (size_t)K * (16 + 8 * (size_t)K);
    if (S == 0) {
        out.put("[]", 2);
    } else {
        out.put("[\n", 2);
        for (int64_t s = 0; s < S; ++s) {
            const char* rec = static_cast<const char*>(records) + (size_t)s * rec_bytes;
            const double* mse = reinterpret_cast<const double*>(rec);
            const int32_t* leaf = reinterpret_cast<const int32_t*>(rec + 8 * (size_t)K);
            const int32_t* sel = reinterpret_cast<const int32_t*>(rec + 16 * (size_t)K);
            // worst case of one signal entry: K results x (K atoms x (~60 + tail) + ~200)
            char* p = out.room(512 + (size_t)K * (256 + 24 * (size_t)K + (size_t)K * (96 + 3 * 32 + 8 * 4 * 7)));
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
                int j = leaf[2 * (k - 1)];                            // the '<=' (last minimum) leaf
                for (int e = 0; e < k; ++e) {                         // base-c digits, least significant first, + 1
                    if (e) *p++ = '-';
                    p = put_int(p, j % connections + 1);
                    j /= connections;
                }
                memcpy(p, atoms_key.data(), atoms_key.size()); p += atoms_key.size();
                for (int e = 0; e < k; ++e) {
                    const int32_t pos = sel[((size_t)(k - 1) * K + e) * 2], atom = sel[((size_t)(k - 1) * K + e) * 2 + 1];
                    if (atom < 0 || atom >= n_atoms) { fclose(f); JFAIL(-3, "record %lld: atom index %d outside the dictionary", (long long)s, atom); }
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
            if (s + 1 < S) { *p++ = ','; *p++ = '\n'; }
            out.advance(p);
        }
        out.put("\n", 1);
        out.put(i0);
        out.put("]", 1);
    }
    if (close_object) {                                               // the enclosing dict(s) of the caller's header
        for (int lv = indent_level - 1; lv >= 0; --lv) {
            out.put("\n", 1);
            out.put(std::string(4 * (size_t)lv, ' '));
            out.put("}", 1);
        }
    }
    out.flush();
    const bool ok = out.ok && fclose(f) == 0;
    if (!ok) JFAIL(-4, "csc_sparvar_json_write: write to %s failed", path);
    return 0;
#undef JFAIL
}

}  // extern "C"

// encoding/gob reader and writer for the one value the reference saves: a frameJob (render.go:33-36, Frame int and
// Bodies []body with the exported fields of physics.go:91-96), the content of the `%010d.data` files of main.go:199-248.
//
// Written from the encoding/gob package documentation; no Go toolchain exists where this was written, so the files
// have not been exchanged with the reference (parity unpinned).  The reader goes by the type descriptors in the
// stream (field names), not by the ids an encoder happens to choose; it understands struct, slice and the scalar
// kinds this value needs and skips unknown struct fields of those kinds.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace gob {

struct Body { uint64_t ID = 0; double Mass = 0, Radius = 0, X = 0, Y = 0, Z = 0, Vx = 0, Vy = 0, Vz = 0; };
struct FrameJob { int64_t Frame = 0; std::vector<Body> Bodies; };

enum { T_BOOL = 1, T_INT = 2, T_UINT = 3, T_FLOAT = 4, T_BYTES = 5, T_STRING = 6, FIRST_USER_ID = 65 };

// ---- writer ----
inline void put_uint(std::string& o, uint64_t u)
{
    if (u < 128) { o.push_back((char)u); return; }
    unsigned char raw[8];
    int n = 0;
    for (uint64_t v = u; v; v >>= 8) n++;
    for (int k = 0; k < n; k++) raw[k] = (unsigned char)(u >> (8 * (n - 1 - k)));
    o.push_back((char)(256 - n));
    o.append((const char*)raw, (size_t)n);
}
inline void put_int(std::string& o, int64_t i) { put_uint(o, i < 0 ? ((uint64_t)(~i) << 1) | 1u : (uint64_t)i << 1); }
inline void put_float(std::string& o, double f)
{
    uint64_t bits, rev = 0;
    std::memcpy(&bits, &f, 8);
    for (int k = 0; k < 8; k++) rev = (rev << 8) | ((bits >> (8 * k)) & 0xff);   // Float64bits, byte-reversed
    put_uint(o, rev);
}
inline void put_string(std::string& o, const std::string& s) { put_uint(o, s.size()); o += s; }
inline std::string message(int64_t tid, const std::string& payload)
{
    std::string m, out;
    put_int(m, tid);
    m += payload;
    put_uint(out, m.size());
    return out + m;
}
inline std::string common(const std::string& name, int64_t tid)
{
    std::string o = "\x01";
    put_string(o, name);
    o += '\x01';
    put_int(o, tid);
    o.push_back('\0');
    return o;
}
inline std::string struct_def(const std::string& name, int64_t tid, const std::vector<std::pair<std::string, int64_t>>& fields)
{
    std::string o = "\x03\x01";                       // wireType.StructT (field 2); structType.CommonType (field 0)
    o += common(name, tid);
    o += '\x01';
    put_uint(o, fields.size());
    for (const auto& f : fields) {
        o += '\x01'; put_string(o, f.first);
        o += '\x01'; put_int(o, f.second);
        o.push_back('\0');
    }
    o.push_back('\0'); o.push_back('\0');
    return o;
}
inline std::string slice_def(const std::string& name, int64_t tid, int64_t elem)
{
    std::string o = "\x02\x01";                       // wireType.SliceT (field 1); sliceType.CommonType
    o += common(name, tid);
    o += '\x01'; put_int(o, elem);
    o.push_back('\0'); o.push_back('\0');
    return o;
}

// gob.NewEncoder(w).Encode(frameJob) (main.go:218-219): descriptors outermost first, zero fields not sent.
inline std::string encode(const FrameJob& job)
{
    const int64_t t_job = FIRST_USER_ID, t_body = FIRST_USER_ID + 1, t_slice = FIRST_USER_ID + 2;
    std::string out = message(-t_job, struct_def("frameJob", t_job, {{"Frame", T_INT}, {"Bodies", t_slice}}));
    out += message(-t_slice, slice_def("[]main.body", t_slice, t_body));
    out += message(-t_body, struct_def("body", t_body, {{"ID", T_UINT}, {"Mass", T_FLOAT}, {"Radius", T_FLOAT}, {"X", T_FLOAT},
                                                         {"Y", T_FLOAT}, {"Z", T_FLOAT}, {"Vx", T_FLOAT}, {"Vy", T_FLOAT},
                                                         {"Vz", T_FLOAT}}));
    std::string v;
    int last = -1;
    if (job.Frame != 0) { put_uint(v, 1); put_int(v, job.Frame); last = 0; }
    if (!job.Bodies.empty()) {
        put_uint(v, (uint64_t)(1 - last));
        put_uint(v, job.Bodies.size());
        for (const Body& b : job.Bodies) {
            int lb = -1;
            if (b.ID != 0) { put_uint(v, 1); put_uint(v, b.ID); lb = 0; }
            const double f[8] = {b.Mass, b.Radius, b.X, b.Y, b.Z, b.Vx, b.Vy, b.Vz};
            for (int k = 0; k < 8; k++)
                if (f[k] != 0) { put_uint(v, (uint64_t)(k + 1 - lb)); put_float(v, f[k]); lb = k + 1; }
            v.push_back('\0');
        }
    }
    v.push_back('\0');
    return out + message(t_job, v);
}

// ---- reader ----
struct Reader {
    const unsigned char* d; size_t n, p = 0;
    Reader(const void* data, size_t len) : d((const unsigned char*)data), n(len) {}
    bool eof() const { return p >= n; }
    uint64_t uint()
    {
        if (p >= n) throw std::runtime_error("gob: truncated");
        const unsigned b = d[p++];
        if (b < 128) return b;
        const unsigned k = 256 - b;
        if (k > 8 || p + k > n) throw std::runtime_error("gob: bad unsigned integer");
        uint64_t v = 0;
        for (unsigned i = 0; i < k; i++) v = (v << 8) | d[p++];
        return v;
    }
    int64_t sint() { const uint64_t u = uint(); return (u & 1) ? ~(int64_t)(u >> 1) : (int64_t)(u >> 1); }
    double flt()
    {
        const uint64_t rev = uint();
        uint64_t bits = 0;
        for (int k = 0; k < 8; k++) bits = (bits << 8) | ((rev >> (8 * k)) & 0xff);
        double f; std::memcpy(&f, &bits, 8);
        return f;
    }
    std::string str()
    {
        const uint64_t k = uint();
        if (p + k > n) throw std::runtime_error("gob: truncated");
        std::string s((const char*)d + p, (size_t)k);
        p += (size_t)k;
        return s;
    }
};

struct WireType { int kind = 0; std::vector<std::pair<std::string, int64_t>> fields; int64_t elem = 0; };   // 1 struct, 2 slice

inline void skip_common(Reader& r)
{
    int f = -1;
    for (;;) {
        const uint64_t d = r.uint();
        if (!d) return;
        f += (int)d;
        if (f == 0) r.str(); else if (f == 1) r.sint(); else throw std::runtime_error("gob: unknown CommonType field");
    }
}

inline WireType read_wire_type(Reader& r)
{
    WireType w;
    int fld = -1;
    for (;;) {
        const uint64_t d = r.uint();
        if (!d) break;
        fld += (int)d;
        int inner = -1;
        if (fld == 2) {                                // StructT
            w.kind = 1;
            for (;;) {
                const uint64_t d2 = r.uint();
                if (!d2) break;
                inner += (int)d2;
                if (inner == 0) skip_common(r);
                else if (inner == 1) {
                    const uint64_t cnt = r.uint();
                    for (uint64_t k = 0; k < cnt; k++) {
                        std::string name; int64_t tid = 0; int f = -1;
                        for (;;) {
                            const uint64_t d3 = r.uint();
                            if (!d3) break;
                            f += (int)d3;
                            if (f == 0) name = r.str(); else if (f == 1) tid = r.sint();
                        }
                        w.fields.emplace_back(name, tid);
                    }
                } else throw std::runtime_error("gob: unknown structType field");
            }
        } else if (fld == 1) {                         // SliceT
            w.kind = 2;
            for (;;) {
                const uint64_t d2 = r.uint();
                if (!d2) break;
                inner += (int)d2;
                if (inner == 0) skip_common(r); else if (inner == 1) w.elem = r.sint();
            }
        } else throw std::runtime_error("gob: unsupported type descriptor");
    }
    return w;
}

inline void skip_value(Reader& r, int64_t tid, const std::map<int64_t, WireType>& types);

inline void read_body(Reader& r, const WireType& t, const std::map<int64_t, WireType>& types, Body& b)
{
    int fld = -1;
    for (;;) {
        const uint64_t d = r.uint();
        if (!d) return;
        fld += (int)d;
        if (fld >= (int)t.fields.size()) throw std::runtime_error("gob: field number out of range");
        const std::string& name = t.fields[(size_t)fld].first;
        const int64_t tid = t.fields[(size_t)fld].second;
        double* dst = name == "Mass" ? &b.Mass : name == "Radius" ? &b.Radius : name == "X" ? &b.X : name == "Y" ? &b.Y
                    : name == "Z" ? &b.Z : name == "Vx" ? &b.Vx : name == "Vy" ? &b.Vy : name == "Vz" ? &b.Vz : nullptr;
        if (name == "ID" && tid == T_UINT) b.ID = r.uint();
        else if (dst && tid == T_FLOAT) *dst = r.flt();
        else skip_value(r, tid, types);
    }
}

inline void skip_value(Reader& r, int64_t tid, const std::map<int64_t, WireType>& types)
{
    if (tid == T_BOOL || tid == T_UINT || tid == T_INT || tid == T_FLOAT) { r.uint(); return; }
    if (tid == T_BYTES || tid == T_STRING) { r.str(); return; }
    auto it = types.find(tid);
    if (it == types.end()) throw std::runtime_error("gob: value of undefined type");
    if (it->second.kind == 2) { const uint64_t c = r.uint(); for (uint64_t k = 0; k < c; k++) skip_value(r, it->second.elem, types); return; }
    int fld = -1;
    for (;;) {
        const uint64_t d = r.uint();
        if (!d) return;
        fld += (int)d;
        if (fld >= (int)it->second.fields.size()) throw std::runtime_error("gob: field number out of range");
        skip_value(r, it->second.fields[(size_t)fld].second, types);
    }
}

// gob.NewDecoder(r).Decode(&frame) (main.go:233-238).
inline FrameJob decode(const void* data, size_t len)
{
    Reader r(data, len);
    std::map<int64_t, WireType> types;
    while (!r.eof()) {
        const uint64_t n = r.uint();
        if (r.p + n > r.n) throw std::runtime_error("gob: truncated message");
        Reader m(r.d + r.p, (size_t)n);
        r.p += (size_t)n;
        const int64_t tid = m.sint();
        if (tid < 0) { types[-tid] = read_wire_type(m); continue; }
        auto it = types.find(tid);
        if (it == types.end() || it->second.kind != 1) throw std::runtime_error("gob: not a frameJob");
        FrameJob job;
        bool saw_bodies = false;
        for (const auto& f : it->second.fields) saw_bodies = saw_bodies || f.first == "Bodies";
        if (!saw_bodies) throw std::runtime_error("gob: not a frameJob");
        int fld = -1;
        for (;;) {
            const uint64_t d = m.uint();
            if (!d) break;
            fld += (int)d;
            if (fld >= (int)it->second.fields.size()) throw std::runtime_error("gob: field number out of range");
            const std::string& name = it->second.fields[(size_t)fld].first;
            const int64_t ftid = it->second.fields[(size_t)fld].second;
            auto st = types.find(ftid);
            if (name == "Frame" && ftid == T_INT) job.Frame = m.sint();
            else if (name == "Bodies" && st != types.end() && st->second.kind == 2) {
                auto bt = types.find(st->second.elem);
                if (bt == types.end() || bt->second.kind != 1) throw std::runtime_error("gob: Bodies is not a slice of structs");
                const uint64_t cnt = m.uint();
                job.Bodies.resize((size_t)cnt);
                for (uint64_t k = 0; k < cnt; k++) read_body(m, bt->second, types, job.Bodies[(size_t)k]);
            } else skip_value(m, ftid, types);
        }
        return job;
    }
    throw std::runtime_error("gob: no value in stream");
}

}  // namespace gob

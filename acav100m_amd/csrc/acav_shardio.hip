// acav_shardio.hip -- native reader of the clustering stage's FEATURE SHARDS (host code only; no kernel in this file).
//
// A feature shard is what the upstream extractor pickles and what the reference reads with `pickle.load`
// (/root/reference/clustering/code/data/clustering.py:172, utils.py load_pickle): a LIST of per-clip DICTS
//     {'filename': str, 'shard_name': str, 'shard_size': int, ...,
//      'audio_features' / 'video_features': [ {'model_key', 'extractor_name', 'dataset',
//                                              'array': {layer: ndarray} | [ndarray, ...] | ndarray}, ... ]}
// which `collate_features` (clustering.py:78-113) turns into one [rows, d] matrix per (model, layer).  Unpickling builds
// ~10 Python objects per clip and view and then copies every vector twice more (np.stack, the upload table): 24 ms per
// 1000-clip shard in each of 40 worker processes on the GPU box = 0.5-0.9 M rows/s against 3-4 M rows/s of GPU training.
// Here the shard file is read, its pickle opcodes are walked ONCE into a small node arena (strings and array payloads
// stay where they are in the file), and every vector is copied straight from the file into its row of the caller's
// destination matrix -- one pass over the bytes, no per-row Python objects, no GIL (the Python side calls this from
// plain threads).
//
// Scope: binary pickle protocols 2-5 as CPython writes them for lists / dicts / tuples / str / bytes / int / float / bool /
// None and numpy ndarrays (`numpy.core.multiarray._reconstruct` + BUILD, or protocol 5's `_frombuffer`; numpy 1.x and 2.x
// module names; protocol 2 is parsed but Python 3 writes array bytes as latin-1 TEXT there: refused).  A vector must be
// a C-contiguous little-endian float32 array with exactly one non-unit dimension.  ANYTHING else -- a text opcode, another
// dtype, rows whose view lists differ, keys missing -- returns ACAV_EUNSUPPORTED and the caller reads that shard with
// `pickle.load` as before: never a different result, only a slower shard.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <thread>

#include "acav_common.h"

namespace {

enum NodeKind : uint8_t { N_NONE, N_BOOL, N_INT, N_FLOAT, N_STR, N_BYTES, N_LIST, N_TUPLE, N_DICT, N_GLOBAL, N_REDUCE, N_NDARRAY, N_DTYPE, N_OPAQUE, N_MARK };

// children of a container node: a range of the arena's int pool (no allocation per node: a 1000-clip shard has ~15 k containers,
// and 40 threads allocating and freeing them ran at the speed of a few)
struct Items {
    std::vector<int> *pool = nullptr;
    size_t start = 0, count = 0;
    size_t size() const { return count; }
    bool empty() const { return count == 0; }
    int operator[](size_t q) const { return (*pool)[start + q]; }
    const int *begin() const { return count ? pool->data() + start : nullptr; }
    const int *end() const { return count ? pool->data() + start + count : nullptr; }
    void clear() { count = 0; }
    void append(const int *v, size_t n)  // v does not point into the pool
    {
        if (count && start + count != pool->size()) {  // not at the end of the pool: move there first
            const size_t ns = pool->size();
            pool->resize(ns + count);
            memmove(pool->data() + ns, pool->data() + start, count * sizeof(int));
            start = ns;
        } else if (!count) {
            start = pool->size();
        }
        pool->insert(pool->end(), v, v + n);
        count += n;
    }
};

struct Node {
    NodeKind kind = N_NONE;
    bool flag = false;        // N_BOOL value; N_DTYPE: float32 little-endian; N_NDARRAY: usable vector
    int64_t i = 0;            // N_INT value; N_STR / N_BYTES / N_NDARRAY: byte offset of the payload in the file
    int64_t len = 0;          // N_STR / N_BYTES: bytes; N_NDARRAY: elements
    int a = -1, b = -1;       // N_GLOBAL: module / name string nodes; N_REDUCE: callable / args
    Items items;              // N_LIST / N_TUPLE: children; N_DICT: key, value, key, value ...
};

struct Arena {  // parse scratch, reused from shard to shard by a worker
    std::vector<Node> nodes;
    std::vector<int> pool;
};

struct View {
    int kind;                 // 0 audio, 1 video
    std::string model_key, layer, extractor, dataset;
    bool has_extractor = false, has_dataset = false;
    int64_t d = 0;
    std::vector<int64_t> off; // payload offset per row
};

}  // namespace

struct acav_pkl_shard {
    const unsigned char *base = nullptr;  // the file's bytes while they are needed (parse, copies)
    size_t size = 0;
    std::vector<unsigned char> own;       // ... owned by the handle (acav_pkl_shard_open) or by a worker of acav_pkl_load_group
    Arena *ar = nullptr;                  // while the shard is being parsed
    int64_t rows = 0;
    std::vector<View> views;
    std::string filenames;            // '\n'-joined
    std::vector<int64_t> name_id;     // per row: identity of the shard_name object (payload offset), -1 = key missing
    std::string names;                // the distinct shard names, '\n'-joined in order of first appearance
    std::vector<int64_t> name_ids;    // their identities, same order
    std::vector<int64_t> shard_size;  // per row, INT64_MIN = key missing
    bool assign = false;              // an ASSIGNMENT shard (the clustering stage's output): integer labels instead of vectors
    std::vector<int64_t> labels;      // [rows][views]
    std::string reason;               // why the shard is unsupported
};

namespace {

struct Parser {
    acav_pkl_shard &S;
    const unsigned char *p, *end;
    std::vector<int> stack, memo, scratch;
    explicit Parser(acav_pkl_shard &s) : S(s), p(s.base), end(s.base + s.size) {}

    int add(NodeKind k)
    {
        S.ar->nodes.emplace_back();
        S.ar->nodes.back().kind = k;
        S.ar->nodes.back().items.pool = &S.ar->pool;
        return (int)S.ar->nodes.size() - 1;
    }
    bool fail(const char *why)
    {
        if (S.reason.empty()) S.reason = why;
        return false;
    }
    bool need(size_t n) { return (size_t)(end - p) >= n; }
    uint64_t rd(int n)
    {
        uint64_t v = 0;
        for (int q = 0; q < n; ++q) v |= (uint64_t)p[q] << (8 * q);
        p += n;
        return v;
    }
    bool pop(int &out)
    {
        if (stack.empty() || S.ar->nodes[(size_t)stack.back()].kind == N_MARK) return fail("stack underflow");
        out = stack.back();
        stack.pop_back();
        return true;
    }
    bool pop_to_mark(std::vector<int> &out)
    {
        size_t m = stack.size();
        while (m > 0 && S.ar->nodes[(size_t)stack[m - 1]].kind != N_MARK) --m;
        if (m == 0) return fail("no mark");
        out.assign(stack.begin() + (long)m, stack.end());
        stack.resize(m - 1);
        return true;
    }
    bool payload(NodeKind k, int lenbytes)
    {
        if (!need((size_t)lenbytes)) return fail("truncated");
        const uint64_t n = rd(lenbytes);
        if (!need(n)) return fail("truncated");
        const int id = add(k);
        S.ar->nodes[(size_t)id].i = (int64_t)(p - S.base);
        S.ar->nodes[(size_t)id].len = (int64_t)n;
        p += n;
        stack.push_back(id);
        return true;
    }
    bool str_is(int id, const char *s) const
    {
        const Node &n = S.ar->nodes[(size_t)id];
        const size_t l = strlen(s);
        return n.kind == N_STR && (size_t)n.len == l && memcmp(S.base + n.i, s, l) == 0;
    }
    bool global_is(int id, const char *mod_a, const char *mod_b, const char *name) const
    {
        const Node &n = S.ar->nodes[(size_t)id];
        return n.kind == N_GLOBAL && (str_is(n.a, mod_a) || (mod_b && str_is(n.a, mod_b))) && str_is(n.b, name);
    }
    // BUILD: obj.__setstate__(state).  The two numpy cases are folded into N_DTYPE / N_NDARRAY; everything else is opaque.
    void build(int obj, int state)
    {
        Node &o = S.ar->nodes[(size_t)obj];
        if (o.kind != N_REDUCE) {
            o.kind = N_OPAQUE;
            return;
        }
        const Node &st = S.ar->nodes[(size_t)state];
        if (global_is(o.a, "numpy", nullptr, "dtype")) {
            // dtype('f4', False, True) + state (3, '<', None, None, None, -1, -1, 0)
            const Node &args = S.ar->nodes[(size_t)o.b];
            const bool le = args.kind == N_TUPLE && args.items.size() >= 1 && st.kind == N_TUPLE && st.items.size() >= 2 && str_is(st.items[1], "<");
            const bool f4 = le && str_is(args.items[0], "f4"), i8 = le && str_is(args.items[0], "i8");
            o.kind = N_DTYPE;
            o.flag = f4;          // the vectors' dtype
            o.len = i8 ? 8 : 0;   // numpy.int64 scalars (assignment shards)
            o.items.clear();
            return;
        }
        if (global_is(o.a, "numpy.core.multiarray", "numpy._core.multiarray", "_reconstruct")) {
            // state (version, shape, dtype, is_fortran, rawdata)
            o.kind = N_NDARRAY;
            o.flag = false;
            if (st.kind != N_TUPLE || st.items.size() != 5) return;
            const Node &shape = S.ar->nodes[(size_t)st.items[1]], &dt = S.ar->nodes[(size_t)st.items[2]], &fo = S.ar->nodes[(size_t)st.items[3]],
                       &raw = S.ar->nodes[(size_t)st.items[4]];
            if (shape.kind != N_TUPLE || shape.items.empty() || dt.kind != N_DTYPE || !dt.flag || raw.kind != N_BYTES) return;
            int64_t elems = 1, nonunit = 0;
            for (int s : shape.items) {
                const Node &dim = S.ar->nodes[(size_t)s];
                if (dim.kind != N_INT || dim.i < 1) return;
                elems *= dim.i;
                nonunit += dim.i != 1;
            }
            // one dimension only: np.stack of (d,) vectors gives [rows, d] (anything else keeps its extra axes in the reference)
            if (shape.items.size() != 1 || raw.len != elems * 4) return;
            if (fo.kind == N_BOOL && fo.flag && nonunit > 1) return;
            o.i = raw.i;
            o.len = elems;
            o.flag = true;
            return;
        }
        o.kind = N_OPAQUE;
    }

    // protocol 5: numpy pickles an array as _frombuffer(buffer, dtype, shape, order) -- no BUILD follows
    void frombuffer(int obj)
    {
        Node &o = S.ar->nodes[(size_t)obj];
        if (global_is(o.a, "numpy.core.multiarray", "numpy._core.multiarray", "scalar")) {
            // numpy.int64(label) of an assignment shard: scalar(dtype('i8'), 8 bytes) -> a plain integer
            const Node &args = S.ar->nodes[(size_t)o.b];
            if (args.kind != N_TUPLE || args.items.size() != 2) return;
            const Node &dt = S.ar->nodes[(size_t)args.items[0]], &raw = S.ar->nodes[(size_t)args.items[1]];
            if (dt.kind != N_DTYPE || dt.len != 8 || raw.kind != N_BYTES || raw.len != 8) return;
            int64_t v;
            memcpy(&v, S.base + raw.i, 8);
            o.kind = N_INT;
            o.i = v;
            return;
        }
        if (!global_is(o.a, "numpy.core.numeric", "numpy._core.numeric", "_frombuffer")) return;
        o.kind = N_NDARRAY;
        o.flag = false;
        const Node &args = S.ar->nodes[(size_t)o.b];
        if (args.kind != N_TUPLE || args.items.size() != 4) return;
        const Node &raw = S.ar->nodes[(size_t)args.items[0]], &dt = S.ar->nodes[(size_t)args.items[1]], &shape = S.ar->nodes[(size_t)args.items[2]];
        if (raw.kind != N_BYTES || dt.kind != N_DTYPE || !dt.flag || shape.kind != N_TUPLE || shape.items.size() != 1) return;
        const Node &dim = S.ar->nodes[(size_t)shape.items[0]];
        if (dim.kind != N_INT || dim.i < 1 || raw.len != dim.i * 4) return;
        o.i = raw.i;
        o.len = dim.i;
        o.flag = true;
    }

    bool run(int &root)
    {
        while (p < end) {
            const unsigned char op = *p++;
            switch (op) {
            case 0x80:  // PROTO
                if (!need(1)) return fail("truncated");
                if (*p < 2 || *p > 5) return fail("pickle protocol outside 2..5");
                ++p;
                break;
            case 0x95:  // FRAME
                if (!need(8)) return fail("truncated");
                p += 8;
                break;
            case '.': {  // STOP
                return pop(root);
            }
            case '(': stack.push_back(add(N_MARK)); break;
            case 'N': stack.push_back(add(N_NONE)); break;
            case 0x88:
            case 0x89: {
                const int id = add(N_BOOL);
                S.ar->nodes[(size_t)id].flag = op == 0x88;
                stack.push_back(id);
                break;
            }
            case 'J':
            case 'K':
            case 'M': {
                const int n = op == 'J' ? 4 : op == 'K' ? 1 : 2;
                if (!need((size_t)n)) return fail("truncated");
                const uint64_t v = rd(n);
                const int id = add(N_INT);
                S.ar->nodes[(size_t)id].i = op == 'J' ? (int64_t)(int32_t)(uint32_t)v : (int64_t)v;
                stack.push_back(id);
                break;
            }
            case 0x8a: {  // LONG1
                if (!need(1)) return fail("truncated");
                const int n = *p++;
                if (n > 8) return fail("integer wider than 64 bits");
                if (!need((size_t)n)) return fail("truncated");
                uint64_t v = rd(n);
                if (n > 0 && n < 8 && (v >> (8 * n - 1)) & 1) v |= ~0ull << (8 * n);  // sign extension
                const int id = add(N_INT);
                S.ar->nodes[(size_t)id].i = (int64_t)v;
                stack.push_back(id);
                break;
            }
            case 'G': {
                if (!need(8)) return fail("truncated");
                p += 8;
                stack.push_back(add(N_FLOAT));
                break;
            }
            case 'X': if (!payload(N_STR, 4)) return false; break;
            case 0x8c: if (!payload(N_STR, 1)) return false; break;
            case 0x8d: if (!payload(N_STR, 8)) return false; break;
            case 'B': if (!payload(N_BYTES, 4)) return false; break;
            case 'C': if (!payload(N_BYTES, 1)) return false; break;
            case 0x8e: if (!payload(N_BYTES, 8)) return false; break;
            case 0x96: if (!payload(N_BYTES, 8)) return false; break;  // BYTEARRAY8 (protocol 5: a writable array's buffer, in band)
            case ']': stack.push_back(add(N_LIST)); break;
            case '}': stack.push_back(add(N_DICT)); break;
            case ')': stack.push_back(add(N_TUPLE)); break;
            case 0x85:
            case 0x86:
            case 0x87: {
                const int n = op - 0x84;
                const int id = add(N_TUPLE);
                int it[3];
                for (int q = n - 1; q >= 0; --q)
                    if (!pop(it[q])) return false;
                S.ar->nodes[(size_t)id].items.append(it, (size_t)n);
                stack.push_back(id);
                break;
            }
            case 't': {
                std::vector<int> &it = scratch;
                if (!pop_to_mark(it)) return false;
                const int id = add(N_TUPLE);
                S.ar->nodes[(size_t)id].items.append(it.data(), it.size());
                stack.push_back(id);
                break;
            }
            case 'a': {
                int v, l;
                if (!pop(v) || stack.empty()) return fail("stack underflow");
                l = stack.back();
                if (S.ar->nodes[(size_t)l].kind != N_LIST) return fail("APPEND to a non-list");
                S.ar->nodes[(size_t)l].items.append(&v, 1);
                break;
            }
            case 'e': {
                std::vector<int> &it = scratch;
                if (!pop_to_mark(it) || stack.empty()) return fail("stack underflow");
                Node &l = S.ar->nodes[(size_t)stack.back()];
                if (l.kind != N_LIST) return fail("APPENDS to a non-list");
                l.items.append(it.data(), it.size());
                break;
            }
            case 's': {
                int v, k;
                if (!pop(v) || !pop(k) || stack.empty()) return fail("stack underflow");
                Node &dct = S.ar->nodes[(size_t)stack.back()];
                if (dct.kind != N_DICT) return fail("SETITEM on a non-dict");
                const int kv[2] = {k, v};
                dct.items.append(kv, 2);
                break;
            }
            case 'u': {
                std::vector<int> &it = scratch;
                if (!pop_to_mark(it) || stack.empty() || (it.size() & 1)) return fail("stack underflow");
                Node &dct = S.ar->nodes[(size_t)stack.back()];
                if (dct.kind != N_DICT) return fail("SETITEMS on a non-dict");
                dct.items.append(it.data(), it.size());
                break;
            }
            case 0x94: {  // MEMOIZE
                if (stack.empty()) return fail("stack underflow");
                memo.push_back(stack.back());
                break;
            }
            case 'q':
            case 'r': {  // BINPUT / LONG_BINPUT
                const int n = op == 'q' ? 1 : 4;
                if (!need((size_t)n) || stack.empty()) return fail("truncated");
                const size_t idx = (size_t)rd(n);
                if (idx > (1u << 28)) return fail("memo index");
                if (memo.size() <= idx) memo.resize(idx + 1, -1);
                memo[idx] = stack.back();
                break;
            }
            case 'h':
            case 'j': {  // BINGET / LONG_BINGET
                const int n = op == 'h' ? 1 : 4;
                if (!need((size_t)n)) return fail("truncated");
                const size_t idx = (size_t)rd(n);
                if (idx >= memo.size() || memo[idx] < 0) return fail("memo miss");
                stack.push_back(memo[idx]);
                break;
            }
            case 'c': {  // GLOBAL: "module\nname\n"
                int ids[2];
                for (int q = 0; q < 2; ++q) {
                    const unsigned char *e = (const unsigned char *)memchr(p, '\n', (size_t)(end - p));
                    if (!e) return fail("truncated");
                    ids[q] = add(N_STR);
                    S.ar->nodes[(size_t)ids[q]].i = (int64_t)(p - S.base);
                    S.ar->nodes[(size_t)ids[q]].len = (int64_t)(e - p);
                    p = e + 1;
                }
                const int id = add(N_GLOBAL);
                S.ar->nodes[(size_t)id].a = ids[0];
                S.ar->nodes[(size_t)id].b = ids[1];
                stack.push_back(id);
                break;
            }
            case 0x93: {  // STACK_GLOBAL
                int name, mod;
                if (!pop(name) || !pop(mod)) return false;
                const int id = add(N_GLOBAL);
                S.ar->nodes[(size_t)id].a = mod;
                S.ar->nodes[(size_t)id].b = name;
                stack.push_back(id);
                break;
            }
            case 'R': {
                int args, fn;
                if (!pop(args) || !pop(fn)) return false;
                const int id = add(N_REDUCE);
                S.ar->nodes[(size_t)id].a = fn;
                S.ar->nodes[(size_t)id].b = args;
                stack.push_back(id);
                frombuffer(id);
                break;
            }
            case 'b': {
                int state;
                if (!pop(state) || stack.empty()) return fail("stack underflow");
                build(stack.back(), state);
                break;
            }
            case 0x81: {  // NEWOBJ: cls.__new__(cls, *args) -- nothing the shard layout needs
                int args, cls;
                if (!pop(args) || !pop(cls)) return false;
                stack.push_back(add(N_OPAQUE));
                break;
            }
            case '0': {
                int x;
                if (!pop(x)) return false;
                break;
            }
            default: return fail("pickle opcode outside the supported binary subset");
            }
        }
        return fail("no STOP");
    }
};

int dict_get(const acav_pkl_shard &S, const Parser &P, const Node &dct, const char *key)
{
    for (size_t q = 0; q + 1 < dct.items.size(); q += 2)
        if (P.str_is(dct.items[q], key)) return dct.items[q + 1];
    return -1;
}

std::string str_of(const acav_pkl_shard &S, int id)
{
    const Node &n = S.ar->nodes[(size_t)id];
    return std::string(reinterpret_cast<const char *>(S.base + n.i), (size_t)n.len);
}

// the shard's layout as shards.py:_shard_columns_from_rows builds it: views in the order of the FIRST row's walk (audio
// features, then video features; feature list order; layer order), every later row with exactly the same sequence
// S.assign: the same walk over an ASSIGNMENT shard -- what subset_selection/code/dataloader.py:17-69 (format_row /
// format_assignments) reads: 'audio_assignments' / 'video_assignments' entries whose 'array' maps layers to integer labels
bool extract(acav_pkl_shard &S, Parser &P, int root)
{
    const Node &top = S.ar->nodes[(size_t)root];
    if (top.kind != N_LIST) return P.fail("the shard is not a list of rows");
    S.rows = (int64_t)top.items.size();
    S.name_id.assign((size_t)S.rows, -1);
    S.shard_size.assign((size_t)S.rows, INT64_MIN);
    static const char *const feature_keys[2][2] = {{"audio_features", "video_features"}, {"audio_assignments", "video_assignments"}};
    for (int64_t r = 0; r < S.rows; ++r) {
        const Node &row = S.ar->nodes[(size_t)top.items[(size_t)r]];
        if (row.kind != N_DICT) return P.fail("a row is not a dict");
        const int fn = dict_get(S, P, row, "filename");
        if (fn < 0 || S.ar->nodes[(size_t)fn].kind != N_STR) return P.fail("a row without a filename string");
        const Node &f = S.ar->nodes[(size_t)fn];
        if (memchr(S.base + f.i, '\n', (size_t)f.len)) return P.fail("newline in a filename");
        if (r) S.filenames.push_back('\n');
        S.filenames.append(reinterpret_cast<const char *>(S.base + f.i), (size_t)f.len);
        const int sn = dict_get(S, P, row, "shard_name");
        if (sn >= 0) {
            const Node &nm = S.ar->nodes[(size_t)sn];
            if (nm.kind != N_STR || memchr(S.base + nm.i, '\n', (size_t)nm.len)) return P.fail("shard_name is not a plain string");
            S.name_id[(size_t)r] = nm.i;  // one unpickled object per payload: identity = offset
            bool seen = false;
            for (int64_t id : S.name_ids) seen |= id == nm.i;
            if (!seen) {
                if (S.name_ids.size() >= 4096) return P.fail("too many distinct shard_name objects");
                if (!S.name_ids.empty()) S.names.push_back('\n');
                S.names.append(reinterpret_cast<const char *>(S.base + nm.i), (size_t)nm.len);
                S.name_ids.push_back(nm.i);
            }
        }
        if (S.assign && sn < 0) return P.fail("an assignment row without shard_name");
        const int ss = dict_get(S, P, row, "shard_size");
        if (ss >= 0) {
            if (S.ar->nodes[(size_t)ss].kind != N_INT && !S.assign) return P.fail("shard_size is not an int");
            if (S.ar->nodes[(size_t)ss].kind == N_INT) S.shard_size[(size_t)r] = S.ar->nodes[(size_t)ss].i;
        }
        size_t vi = 0;
        for (int kind = 0; kind < 2; ++kind) {
            const int fl = dict_get(S, P, row, feature_keys[S.assign ? 1 : 0][kind]);
            if (fl < 0) continue;
            const Node &feats = S.ar->nodes[(size_t)fl];
            if (feats.kind != N_LIST && feats.kind != N_TUPLE) return P.fail("a feature list is not a list");
            for (int fid : feats.items) {
                const Node &feat = S.ar->nodes[(size_t)fid];
                if (feat.kind != N_DICT) return P.fail("a feature entry is not a dict");
                const int mk = dict_get(S, P, feat, "model_key"), arr = dict_get(S, P, feat, "array");
                if (mk < 0 || S.ar->nodes[(size_t)mk].kind != N_STR || arr < 0) return P.fail("feature entry without model_key / array");
                const int ex = dict_get(S, P, feat, "extractor_name"), ds = dict_get(S, P, feat, "dataset");
                if ((ex >= 0 && S.ar->nodes[(size_t)ex].kind != N_STR && S.ar->nodes[(size_t)ex].kind != N_NONE) ||
                    (ds >= 0 && S.ar->nodes[(size_t)ds].kind != N_STR && S.ar->nodes[(size_t)ds].kind != N_NONE))
                    return P.fail("extractor_name / dataset is not a string");
                const Node &a = S.ar->nodes[(size_t)arr];
                std::vector<std::pair<std::string, int>> layers;
                if (a.kind == N_DICT) {
                    for (size_t q = 0; q + 1 < a.items.size(); q += 2) {
                        if (S.ar->nodes[(size_t)a.items[q]].kind != N_STR) return P.fail("a layer key is not a string");
                        layers.emplace_back(str_of(S, a.items[q]), a.items[q + 1]);
                    }
                } else if (a.kind == N_LIST || a.kind == N_TUPLE) {
                    for (size_t q = 0; q < a.items.size(); ++q) layers.emplace_back("layer_" + std::to_string(q), a.items[q]);
                } else {
                    if (S.assign) return P.fail("scalar assignment arrays are not supported by the reference");
                    layers.emplace_back("model", arr);
                }
                for (auto &lv : layers) {
                    const Node &vec = S.ar->nodes[(size_t)lv.second];
                    if (S.assign ? vec.kind != N_INT : (vec.kind != N_NDARRAY || !vec.flag))
                        return P.fail(S.assign ? "a label is not an integer" : "a feature vector is not a plain float32 ndarray");
                    if (r == 0) {
                        View v;
                        v.kind = kind;
                        v.model_key = str_of(S, mk);
                        v.layer = lv.first;
                        v.has_extractor = ex >= 0 && S.ar->nodes[(size_t)ex].kind == N_STR;
                        v.has_dataset = ds >= 0 && S.ar->nodes[(size_t)ds].kind == N_STR;
                        if (v.has_extractor) v.extractor = str_of(S, ex);
                        if (v.has_dataset) v.dataset = str_of(S, ds);
                        v.d = S.assign ? 1 : vec.len;
                        v.off.reserve((size_t)S.rows);
                        for (const View &o : S.views)  // (an assignment row is keyed by (model key, layer) alone)
                            if ((S.assign || o.kind == v.kind) && o.model_key == v.model_key && o.layer == v.layer)
                                return P.fail("a view appears twice in a row");
                        S.views.push_back(std::move(v));
                    }
                    if (vi >= S.views.size()) return P.fail("rows with different view lists");
                    View &v = S.views[vi];
                    const Node &mkn = S.ar->nodes[(size_t)mk];
                    if (v.kind != kind || v.d != (S.assign ? 1 : vec.len) || (size_t)mkn.len != v.model_key.size() ||
                        memcmp(S.base + mkn.i, v.model_key.data(), v.model_key.size()) != 0 || v.layer != lv.first)
                        return P.fail("rows with different view lists");
                    if (S.assign) S.labels.push_back(vec.i);
                    else v.off.push_back(vec.i);
                    ++vi;
                }
            }
        }
        if (vi != S.views.size()) return P.fail("rows with different view lists");
    }
    return true;
}

}  // namespace

// The whole file by pread into `buf` (a mapping per shard would take the process-wide mm lock in every thread: 40 loader
// threads then run at the speed of 4).  false: cannot be read.
static bool read_file(const char *path, std::vector<unsigned char> &buf, size_t &size)
{
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && st.st_size > 0;
    if (ok) {
        size = (size_t)st.st_size;
        if (buf.size() < size) buf.resize(size);
        size_t got = 0;
        while (ok && got < size) {
            const ssize_t r = pread(fd, buf.data() + got, size - got, (off_t)got);
            if (r <= 0) ok = false;
            else got += (size_t)r;
        }
    }
    close(fd);
    return ok;
}

static bool parse_shard(acav_pkl_shard *S, Arena &ar)
{
    bool ok = false;
    ar.nodes.clear();
    ar.pool.clear();
    S->ar = &ar;
    try {
        Parser P(*S);
        int root = -1;
        ok = P.run(root) && extract(*S, P, root);
    } catch (const std::bad_alloc &) {
        S->reason = "out of memory";
    }
    S->ar = nullptr;
    return ok;
}

// Reads and parses the shard.  ACAV_EUNSUPPORTED: a pickle this reader does not cover (acav_last_error() says why) -- the
// caller falls back to pickle.load; ACAV_EINVAL: the file cannot be read.
ACAV_EXPORT int acav_pkl_shard_open(const char *path, acav_pkl_shard **out)
{
    ACAV_REQUIRE(path && out, ACAV_EINVAL, "acav_pkl_shard_open: null argument");
    *out = nullptr;
    acav_pkl_shard *S = new (std::nothrow) acav_pkl_shard();
    ACAV_REQUIRE(S, ACAV_ENOMEM, "acav_pkl_shard_open: out of memory");
    if (!read_file(path, S->own, S->size)) {
        acav::set_error("acav_pkl_shard_open: cannot read %s", path);
        delete S;
        return ACAV_EINVAL;
    }
    S->base = S->own.data();
    Arena ar;
    if (!parse_shard(S, ar)) {
        acav::set_error("acav_pkl_shard_open: %s: %s", path, S->reason.c_str());
        delete S;
        return ACAV_EUNSUPPORTED;
    }
    *out = S;
    return ACAV_OK;
}

ACAV_EXPORT int acav_pkl_shard_close(acav_pkl_shard *S)
{
    delete S;
    return ACAV_OK;
}

ACAV_EXPORT int acav_pkl_shard_info(const acav_pkl_shard *S, int64_t *rows, int *nviews, int *n_names)
{
    ACAV_REQUIRE(S, ACAV_EINVAL, "acav_pkl_shard_info: null handle");
    if (rows) *rows = S->rows;
    if (nviews) *nviews = (int)S->views.size();
    if (n_names) *n_names = (int)S->name_ids.size();
    return ACAV_OK;
}

// view v: kind (0 audio, 1 video), model key, layer name, extractor_name / dataset of the FIRST row (NULL when absent or None),
// vector length.  The strings live as long as the handle.
ACAV_EXPORT int acav_pkl_shard_view(const acav_pkl_shard *S, int v, int *kind, const char **model_key, const char **layer,
                                    const char **extractor, const char **dataset, int64_t *d)
{
    ACAV_REQUIRE(S && v >= 0 && (size_t)v < S->views.size(), ACAV_EINVAL, "acav_pkl_shard_view: view %d out of range", v);
    const View &w = S->views[(size_t)v];
    if (kind) *kind = w.kind;
    if (model_key) *model_key = w.model_key.c_str();
    if (layer) *layer = w.layer.c_str();
    if (extractor) *extractor = w.has_extractor ? w.extractor.c_str() : nullptr;
    if (dataset) *dataset = w.has_dataset ? w.dataset.c_str() : nullptr;
    if (d) *d = w.d;
    return ACAV_OK;
}

// dst[r * row_stride .. + d) = vector of row r, r = 0 .. rows-1 (row_stride in floats, >= d); dst is host memory
ACAV_EXPORT int acav_pkl_shard_copy_view(const acav_pkl_shard *S, int v, float *dst, int64_t row_stride)
{
    ACAV_REQUIRE(S && dst && v >= 0 && (size_t)v < S->views.size(), ACAV_EINVAL, "acav_pkl_shard_copy_view: bad argument");
    ACAV_REQUIRE(S->base, ACAV_ESTATE, "acav_pkl_shard_copy_view: a handle of acav_pkl_load_group keeps the row metadata only");
    const View &w = S->views[(size_t)v];
    ACAV_REQUIRE(row_stride >= w.d, ACAV_EINVAL, "acav_pkl_shard_copy_view: row stride %lld < d %lld", (long long)row_stride, (long long)w.d);
    for (int64_t r = 0; r < S->rows; ++r) memcpy(dst + r * row_stride, S->base + w.off[(size_t)r], (size_t)w.d * 4);
    return ACAV_OK;
}

// Row metadata.  filenames: the rows' filenames joined by '\n' (UTF-8, *filenames_len bytes, not terminated by a newline).
// names / name_ids: the DISTINCT shard_name objects of the pickle ('\n'-joined, in order of first appearance) and their
// identities; name_of_row[r] = identity of row r's shard_name (-1: the row has no such key) -- rows that unpickle to the SAME
// str object share an identity (the writer of the assignment shards relies on that: pickle memoises by identity).
// shard_size[r]: INT64_MIN when the row has no such key.  All pointers live as long as the handle.
ACAV_EXPORT int acav_pkl_shard_meta(const acav_pkl_shard *S, const char **filenames, int64_t *filenames_len, const char **names,
                                    int64_t *names_len, const int64_t **name_ids, const int64_t **name_of_row,
                                    const int64_t **shard_size)
{
    ACAV_REQUIRE(S, ACAV_EINVAL, "acav_pkl_shard_meta: null handle");
    if (filenames) *filenames = S->filenames.data();
    if (filenames_len) *filenames_len = (int64_t)S->filenames.size();
    if (names) *names = S->names.data();
    if (names_len) *names_len = (int64_t)S->names.size();
    if (name_ids) *name_ids = S->name_ids.data();
    if (name_of_row) *name_of_row = S->name_id.data();
    if (shard_size) *shard_size = S->shard_size.data();
    return ACAV_OK;
}

// A GROUP of shards at once, on `threads` threads of this library (one call from Python: no GIL traffic, no worker
// processes): shard i is read, parsed and its vectors copied into rows [base[i], base[i] + rows_i) of the destination
// matrices dest[v] ([*, dims[v]] float32, host memory), v = the caller's views (kind 0 audio / 1 video, model key, layer).
// status[i]: 0 done -- handles[i] then holds the row metadata (acav_pkl_shard_info / _view / _meta; close it);
//            1 outside the native reader's subset or unreadable (the caller reads it with pickle.load);
//            2 the shard does not fit: more than n_expect[i] rows, another view set, another vector length.
ACAV_EXPORT int acav_pkl_load_group(const char *const *paths, int n, const int64_t *base, const int64_t *n_expect, int nviews,
                                    const int *kinds, const char *const *model_keys, const char *const *layers, const int64_t *dims,
                                    float *const *dest, int threads, acav_pkl_shard **handles, int *status)
{
    ACAV_REQUIRE(paths && base && n_expect && kinds && model_keys && layers && dims && dest && handles && status && n >= 0 && nviews > 0,
                 ACAV_EINVAL, "acav_pkl_load_group: bad argument");
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    std::atomic<int> next(0);
    std::atomic<bool> oom(false);
    auto worker = [&]() {
        std::vector<unsigned char> buf;
        Arena ar;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            handles[i] = nullptr;
            status[i] = 1;
            acav_pkl_shard *S = new (std::nothrow) acav_pkl_shard();
            if (!S) {
                oom = true;
                continue;
            }
            try {
                if (read_file(paths[i], buf, S->size)) {
                    S->base = buf.data();
                    if (parse_shard(S, ar)) {
                        status[i] = 0;
                        if (S->rows > n_expect[i] || (int)S->views.size() != nviews) status[i] = 2;
                        int map[64];
                        for (int v = 0; status[i] == 0 && v < nviews && v < 64; ++v) {
                            map[v] = -1;
                            for (size_t w = 0; w < S->views.size(); ++w)
                                if (S->views[w].kind == kinds[v] && S->views[w].model_key == model_keys[v] && S->views[w].layer == layers[v])
                                    map[v] = (int)w;
                            if (map[v] < 0 || S->views[(size_t)map[v]].d != dims[v]) status[i] = 2;
                        }
                        if (nviews > 64) status[i] = 2;
                        for (int v = 0; status[i] == 0 && v < nviews; ++v) {
                            const View &w = S->views[(size_t)map[v]];
                            float *dst = dest[v] + base[i] * dims[v];
                            for (int64_t r = 0; r < S->rows; ++r) memcpy(dst + r * dims[v], buf.data() + w.off[(size_t)r], (size_t)dims[v] * 4);
                        }
                    }
                }
            } catch (const std::bad_alloc &) {
                oom = true;
                status[i] = 1;
            }
            S->base = nullptr;  // the bytes belong to this worker's buffer
            for (View &w : S->views) std::vector<int64_t>().swap(w.off);
            if (status[i] == 0) handles[i] = S;
            else delete S;
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    } catch (...) {  // fewer threads than asked for: the ones that started (and this one) do the work
    }
    worker();
    for (std::thread &t : pool) t.join();
    ACAV_REQUIRE(!oom.load(), ACAV_ENOMEM, "acav_pkl_load_group: out of memory");
    return ACAV_OK;
}

// ASSIGNMENT shards (the clustering stage's output, the selection stage's input): n shards parsed on `threads` threads.
// status[i] 0: handles[i] holds rows, views ((model key, layer) in the order of the first row's walk: acav_pkl_shard_view,
// d = 1), filenames / shard names (acav_pkl_shard_meta) and the labels (acav_pkl_shard_labels); 1: outside the subset.
ACAV_EXPORT int acav_pkl_assign_load_group(const char *const *paths, int n, int threads, acav_pkl_shard **handles, int *status)
{
    ACAV_REQUIRE(paths && handles && status && n >= 0, ACAV_EINVAL, "acav_pkl_assign_load_group: bad argument");
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    std::atomic<int> next(0);
    std::atomic<bool> oom(false);
    auto worker = [&]() {
        std::vector<unsigned char> buf;
        Arena ar;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            handles[i] = nullptr;
            status[i] = 1;
            acav_pkl_shard *S = new (std::nothrow) acav_pkl_shard();
            if (!S) {
                oom = true;
                continue;
            }
            S->assign = true;
            try {
                if (read_file(paths[i], buf, S->size)) {
                    S->base = buf.data();
                    if (parse_shard(S, ar)) status[i] = 0;
                }
            } catch (const std::bad_alloc &) {
                oom = true;
                status[i] = 1;
            }
            S->base = nullptr;
            if (status[i] == 0) handles[i] = S;
            else delete S;
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    } catch (...) {
    }
    worker();
    for (std::thread &t : pool) t.join();
    ACAV_REQUIRE(!oom.load(), ACAV_ENOMEM, "acav_pkl_assign_load_group: out of memory");
    return ACAV_OK;
}

// labels of an assignment-shard handle: int64 [rows][views], alive as long as the handle
ACAV_EXPORT int acav_pkl_shard_labels(const acav_pkl_shard *S, const int64_t **labels)
{
    ACAV_REQUIRE(S && labels && S->assign, ACAV_EINVAL, "acav_pkl_shard_labels: not an assignment-shard handle");
    *labels = S->labels.data();
    return ACAV_OK;
}

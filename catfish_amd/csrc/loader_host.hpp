// Host side of the ingest (no device work): the file read of the reference's per-file loop (catfish/catfish:50-56 ->
// infer.process_signal, infer.py:77-93) for the read format this image can hold -- there is no HDF5 library here, so a read
// is a one-dimensional C-order little-endian int16 .npy (DAC codes after the leader trim).  Many files are read by a small
// pool of host threads straight into ONE caller-owned buffer (the pinned staging buffer of the streaming pipeline), back
// to back in the order given: the per-file Python loader costs ~15 us a file under the GIL, which is what bounded a rank of
// the CLI (12 500 x 4096-sample files: 0.19 s of loading against 0.16 s of device work).  Included by catfish_hip.hip.
#include <dirent.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cerrno>
#include <cstring>
#include <memory>
#include <thread>
#include <new>
#include <stdexcept>

namespace cf_loader {

struct Item {
    std::vector<unsigned char> bytes;    // whole file (allocated only once the header has vouched for its size)
    size_t data_off = 0;
    int64_t count = -1;                  // samples; < 0: not a plain int16 vector (or unreadable)
};

static const size_t HEAD_FIRST = 4096;          // bytes read before the header has been judged
static const size_t HEAD_MAX = 1u << 20;        // no numpy header of a one-dimensional array is anywhere near this long

// the byte-level header check of catfish_amd/infer.py::_read_npy_int16 on the first `have` bytes of a file of `file_size`
// bytes: magic, version 1 or 2, '<i2', C order, one dimension, and the file exactly as long as header + data.
// -> 1 ok (data_off, count set), 0 not such a file, -1 the header is longer than `have` (need_head set: read that much and call again)
static int parse(const unsigned char* b, size_t have, size_t file_size, size_t& data_off, int64_t& count, size_t& need_head) {
    if (file_size < 12 || have < 12 || memcmp(b, "\x93NUMPY", 6) != 0 || (b[6] != 1 && b[6] != 2)) return 0;
    size_t hlen, off;
    if (b[6] == 1) { hlen = (size_t)b[8] | ((size_t)b[9] << 8); off = 10; }
    else { hlen = (size_t)b[8] | ((size_t)b[9] << 8) | ((size_t)b[10] << 16) | ((size_t)b[11] << 24); off = 12; }
    if (hlen > HEAD_MAX || off + hlen > file_size) return 0;
    if (off + hlen > have) { need_head = off + hlen; return -1; }
    const std::string header(reinterpret_cast<const char*>(b + off), hlen);
    if (header.find("'descr': '<i2'") == std::string::npos || header.find("'fortran_order': False") == std::string::npos) return 0;
    const size_t a = header.find("'shape': (");
    if (a == std::string::npos) return 0;
    const size_t close = header.find(')', a);
    if (close == std::string::npos) return 0;
    // exactly one dimension: digits, optional blanks, one comma
    size_t p = a + 10;
    while (p < close && header[p] == ' ') ++p;
    int64_t n = 0;
    size_t digits = 0;
    while (p < close && header[p] >= '0' && header[p] <= '9') { n = n * 10 + (header[p] - '0'); ++p; ++digits; if (digits > 15) return 0; }
    if (digits == 0) return 0;
    int commas = 0;
    for (; p < close; ++p) {
        if (header[p] == ',') ++commas;
        else if (header[p] != ' ') return 0;
    }
    if (commas > 1) return 0;
    if (file_size != off + hlen + 2 * (size_t)n) return 0;
    data_off = off + hlen;
    count = n;
    return 1;
}

static bool read_exact(int fd, unsigned char* dst, size_t n, size_t at) {
    size_t got = 0;
    while (got < n) {
        const ssize_t r = pread(fd, dst + got, n - got, (off_t)(at + got));
        if (r <= 0) return false;
        got += (size_t)r;
    }
    return true;
}

// Header first: a directory, a FAST5, an .npz or a multi-gigabyte file of another kind is turned away after at most 4 KiB (its
// header length, if it claims to be numpy) -- only a file whose header matches its size is read whole.
static bool slurp(int dirfd, const char* path, Item& it) {       // path relative to dirfd (AT_FDCWD: as open() takes it)
    const int fd = openat(dirfd, path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 12;
    if (ok) {
        const size_t size = (size_t)st.st_size;
        unsigned char first[HEAD_FIRST];
        size_t have = std::min(size, HEAD_FIRST), need = 0;
        ok = read_exact(fd, first, have, 0);
        int verdict = ok ? parse(first, have, size, it.data_off, it.count, need) : 0;
        if (verdict < 0) {                                // a header beyond the first 4 KiB (never seen; bounded by HEAD_MAX)
            std::vector<unsigned char> head(need);
            ok = read_exact(fd, head.data(), need, 0);
            verdict = ok ? parse(head.data(), need, size, it.data_off, it.count, need) : 0;
        }
        ok = ok && verdict == 1;
        if (ok) {
            it.bytes.resize(size);
            if (have == HEAD_FIRST || size == have) memcpy(it.bytes.data(), first, have);      // (else the long-header case: read anew)
            else have = 0;
            ok = read_exact(fd, it.bytes.data() + have, size - have, have);
        }
    }
    close(fd);
    if (!ok) it.count = -1;
    return ok;
}

// body(t) on nt host threads; a thread that cannot be started (std::system_error) must not leave joinable ones behind
template <class F>
static void run_pool(int nt, F body) {
    std::vector<std::thread> pool;
    try {
        for (int t = 0; t < nt; ++t) pool.emplace_back(body, t);
    } catch (...) {
        for (std::thread& th : pool) th.join();
        throw;
    }
    for (std::thread& th : pool) th.join();
}

}  // namespace cf_loader

static int load_npy_int16(int dirfd, const char* paths, const int64_t* path_bounds, int64_t n_files, int16_t* out, int64_t capacity,
                          int64_t* lengths, int64_t* total, int32_t n_threads) {
    if (n_files < 0 || capacity < 0) return fail(CF_ERR_INVALID, "cf_load_npy_int16: negative size");
    if (total) *total = 0;
    if (n_files == 0) return CF_OK;
    if (!paths || !path_bounds || !lengths || (!out && capacity > 0)) return fail(CF_ERR_INVALID, "cf_load_npy_int16: null buffer");
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : 4, std::min<int64_t>(n_files, 64)));
    std::vector<cf_loader::Item> items((size_t)n_files);
    std::vector<int64_t> bad((size_t)nt, -1);                    // first offending file per thread
    std::vector<int64_t> nomem((size_t)nt, -1);                  // first file whose bytes did not fit host memory, per thread
    cf_loader::run_pool(nt, [&](int t) {
        for (int64_t i = t; i < n_files; i += nt) {              // interleaved: neighbouring files are neighbours on disk
            cf_loader::Item& it = items[(size_t)i];
            bool ok = false;
            try {
                ok = cf_loader::slurp(dirfd, paths + path_bounds[i], it);
            } catch (const std::bad_alloc&) {                   // a thread may not throw: remembered, reported as CF_ERR_NOMEM below
                ok = false;
                if (nomem[(size_t)t] < 0) nomem[(size_t)t] = i;
            } catch (...) {
                ok = false;
            }
            if (!ok) {
                it.count = -1;
                if (bad[(size_t)t] < 0) bad[(size_t)t] = i;
            }
        }
    });
    int64_t first_bad = -1, first_nomem = -1;
    for (int64_t b : bad)
        if (b >= 0 && (first_bad < 0 || b < first_bad)) first_bad = b;
    for (int64_t b : nomem)
        if (b >= 0 && (first_nomem < 0 || b < first_nomem)) first_nomem = b;
    // out of memory is not "another kind of file": the caller must not answer it by loading the same bytes again elsewhere
    if (first_nomem >= 0)
        return fail(CF_ERR_NOMEM, std::string("cf_load_npy_int16: out of host memory while reading ") + (paths + path_bounds[first_nomem]));
    if (first_bad >= 0)
        return fail(CF_ERR_INVALID, std::string("cf_load_npy_int16: not a readable one-dimensional little-endian int16 .npy: ") +
                                        (paths + path_bounds[first_bad]));
    std::vector<int64_t> offs((size_t)n_files + 1, 0);
    for (int64_t i = 0; i < n_files; ++i) {
        lengths[i] = items[(size_t)i].count;
        offs[(size_t)i + 1] = offs[(size_t)i] + items[(size_t)i].count;
    }
    if (total) *total = offs[(size_t)n_files];
    if (offs[(size_t)n_files] > capacity) return fail(CF_ERR_INVALID, "cf_load_npy_int16: the reads do not fit the buffer");
    cf_loader::run_pool(nt, [&](int t) {
        for (int64_t i = t; i < n_files; i += nt) {
            const cf_loader::Item& it = items[(size_t)i];
            if (it.count > 0) memcpy(out + offs[(size_t)i], it.bytes.data() + it.data_off, 2 * (size_t)it.count);
        }
    });
    return CF_OK;
}

// (no C++ exception may cross the C ABI: std::bad_alloc from the tables, std::system_error from std::thread)
extern "C" int cf_load_npy_int16(const char* paths, const int64_t* path_bounds, int64_t n_files, int16_t* out, int64_t capacity,
                                 int64_t* lengths, int64_t* total, int32_t n_threads) {
    try {
        return load_npy_int16(AT_FDCWD, paths, path_bounds, n_files, out, capacity, lengths, total, n_threads);
    } catch (const std::bad_alloc&) {
        return fail(CF_ERR_NOMEM, "cf_load_npy_int16: out of host memory");
    } catch (const std::exception& e) {
        return fail(CF_ERR_INVALID, std::string("cf_load_npy_int16: ") + e.what());
    }
}

// Sizes on disk of many entries of ONE directory (the listing of the reference's per-file loop, catfish/catfish:49-50, needs them to
// cut the file list into blocks of equal work and into batches before anything is read): fstatat relative to one directory handle
// from a small pool of threads, each a contiguous block of the names -- 12 500 entries cost a Python loop of os.stat 19 ms, this 3-5.
static int stat_files(const char* dir, const char* names, const int64_t* name_bounds, int64_t n_files, int64_t* sizes, int32_t n_threads) {
    if (n_files < 0) return fail(CF_ERR_INVALID, "cf_stat_files: negative count");
    if (n_files == 0) return CF_OK;
    if (!dir || !names || !name_bounds || !sizes) return fail(CF_ERR_INVALID, "cf_stat_files: null buffer");
    const int dfd = open(dir, O_RDONLY | O_DIRECTORY | O_CLOEXEC);
    if (dfd < 0) return fail(CF_ERR_INVALID, std::string("cf_stat_files: cannot open directory ") + dir + ": " + strerror(errno));
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : 4, std::min<int64_t>((n_files + 255) / 256, 64)));
    std::vector<int64_t> bad((size_t)nt, -1);
    std::vector<int> bad_errno((size_t)nt, 0);
    try {
        cf_loader::run_pool(nt, [&](int t) {
            const int64_t lo = n_files * t / nt, hi = n_files * (t + 1) / nt;
            for (int64_t i = lo; i < hi; ++i) {
                struct stat st;
                if (fstatat(dfd, names + name_bounds[i], &st, 0) == 0) sizes[i] = (int64_t)st.st_size;
                else {
                    sizes[i] = -1;
                    if (bad[(size_t)t] < 0) { bad[(size_t)t] = i; bad_errno[(size_t)t] = errno; }
                }
            }
        });
    } catch (...) {
        close(dfd);
        throw;
    }
    close(dfd);
    for (int t = 0; t < nt; ++t)
        if (bad[(size_t)t] >= 0)
            return fail(CF_ERR_INVALID, std::string("cf_stat_files: ") + dir + "/" + (names + name_bounds[bad[(size_t)t]]) + ": " +
                                            strerror(bad_errno[(size_t)t]));
    return CF_OK;
}

extern "C" int cf_stat_files(const char* dir, const char* names, const int64_t* name_bounds, int64_t n_files, int64_t* sizes, int32_t n_threads) {
    try {
        return stat_files(dir, names, name_bounds, n_files, sizes, n_threads);
    } catch (const std::bad_alloc&) {
        return fail(CF_ERR_NOMEM, "cf_stat_files: out of host memory");
    } catch (const std::exception& e) {
        return fail(CF_ERR_INVALID, std::string("cf_stat_files: ") + e.what());
    }
}

// The listing of the input directory as an object (catfish/catfish:49-50: `input_files = os.listdir(input_dir)`): every rank of a
// sharded job needs the same ORDER of all names, the sizes of one block of them and the names of the block it ends up working on --
// not a Python string per entry of a 100 000-file directory on each of 8 ranks.  Names are read with readdir, ordered bytewise
// (= sorted() of the decoded names whenever they are valid UTF-8, whose byte order is its code-point order) by an 8-byte key behind the
// common prefix with strcmp as the tie-break, and kept as one blob.
struct cf_listing {
    std::string dir;
    std::vector<char> blob;              // names, NUL-terminated, in READ order
    std::vector<uint32_t> at;            // offset of entry i (sorted order) in blob
};

namespace cf_loader {
static uint64_t key_of(const char* s, size_t skip) {
    // the 8 bytes behind the common prefix, big-endian, NUL-padded: integer order = byte order of that stretch
    uint64_t k = 0;
    size_t n = strlen(s);
    for (size_t i = 0; i < 8; ++i) k = (k << 8) | (skip + i < n ? (unsigned char)s[skip + i] : 0u);
    return k;
}
}  // namespace cf_loader

static int listing_open(const char* dir, cf_listing** out, int64_t* n_entries, uint64_t* digest) {
    if (!dir || !out) return fail(CF_ERR_INVALID, "cf_listing_open: null argument");
    *out = nullptr;
    DIR* d = opendir(dir);
    if (!d) return fail(CF_ERR_INVALID, std::string("cf_listing_open: cannot open directory ") + dir + ": " + strerror(errno));
    std::unique_ptr<cf_listing> l(new cf_listing);
    l->dir = dir;
    std::vector<uint32_t> off;
    try {
        for (;;) {
            errno = 0;
            const struct dirent* e = readdir(d);
            if (!e) {
                if (errno != 0) {
                    const int err = errno;
                    closedir(d);
                    return fail(CF_ERR_INVALID, std::string("cf_listing_open: reading ") + dir + ": " + strerror(err));
                }
                break;
            }
            const char* nm = e->d_name;
            if (nm[0] == '.' && (nm[1] == 0 || (nm[1] == '.' && nm[2] == 0))) continue;
            const size_t len = strlen(nm);
            if (l->blob.size() + len + 1 > 0xffffffffull) {
                closedir(d);
                return fail(CF_ERR_INVALID, "cf_listing_open: more than 4 GiB of names");
            }
            off.push_back((uint32_t)l->blob.size());
            l->blob.insert(l->blob.end(), nm, nm + len + 1);
        }
    } catch (...) {
        closedir(d);
        throw;
    }
    closedir(d);
    const char* base = l->blob.data();
    // common prefix of all names ("read_", "channel_12_read_" ...): the key starts behind it
    size_t skip = off.empty() ? 0 : strlen(base + off[0]);
    for (size_t i = 1; i < off.size() && skip > 0; ++i) {
        const char* a = base + off[0];
        const char* b = base + off[i];
        size_t k = 0;
        while (k < skip && a[k] == b[k] && b[k]) ++k;
        skip = k;
    }
    std::vector<std::pair<uint64_t, uint32_t>> keyed(off.size());
    for (size_t i = 0; i < off.size(); ++i) keyed[i] = {cf_loader::key_of(base + off[i], skip), off[i]};
    std::sort(keyed.begin(), keyed.end(), [&](const std::pair<uint64_t, uint32_t>& x, const std::pair<uint64_t, uint32_t>& y) {
        if (x.first != y.first) return x.first < y.first;
        return strcmp(base + x.second, base + y.second) < 0;        // equal keys: equal up to skip + 8 bytes (or both shorter)
    });
    l->at.resize(off.size());
    // two FNV-1a style 64-bit hashes over the sorted names (NUL included): what the ranks compare
    uint64_t h0 = 1469598103934665603ull, h1 = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < keyed.size(); ++i) {
        l->at[i] = keyed[i].second;
        for (const char* c = base + keyed[i].second;; ++c) {
            h0 = (h0 ^ (unsigned char)*c) * 1099511628211ull;
            h1 = (h1 ^ (unsigned char)*c) * 0x100000001b3ull + 0x632be59bd9b4e019ull;
            if (!*c) break;
        }
    }
    if (n_entries) *n_entries = (int64_t)l->at.size();
    if (digest) { digest[0] = h0; digest[1] = h1; }
    *out = l.release();
    return CF_OK;
}

extern "C" int cf_listing_open(const char* dir, cf_listing** out, int64_t* n_entries, uint64_t* digest) {
    try {
        return listing_open(dir, out, n_entries, digest);
    } catch (const std::bad_alloc&) {
        return fail(CF_ERR_NOMEM, "cf_listing_open: out of host memory");
    } catch (const std::exception& e) {
        return fail(CF_ERR_INVALID, std::string("cf_listing_open: ") + e.what());
    }
}

extern "C" void cf_listing_close(cf_listing* l) { delete l; }

static bool listing_range_ok(const cf_listing* l, int64_t lo, int64_t hi) { return l && lo >= 0 && lo <= hi && hi <= (int64_t)l->at.size(); }

extern "C" int cf_listing_sizes(const cf_listing* l, int64_t lo, int64_t hi, int64_t* sizes, int32_t n_threads) {
    if (!listing_range_ok(l, lo, hi)) return fail(CF_ERR_INVALID, "cf_listing_sizes: bad range");
    if (hi == lo) return CF_OK;
    if (!sizes) return fail(CF_ERR_INVALID, "cf_listing_sizes: null buffer");
    try {
        const int64_t n = hi - lo;
        std::vector<int64_t> bounds((size_t)n + 1);
        // stat_files takes names back to back: here they are scattered in the blob, so hand it offsets relative to the blob's start
        for (int64_t i = 0; i < n; ++i) bounds[(size_t)i] = (int64_t)l->at[(size_t)(lo + i)];
        bounds[(size_t)n] = 0;
        return stat_files(l->dir.c_str(), l->blob.data(), bounds.data(), n, sizes, n_threads);
    } catch (const std::bad_alloc&) {
        return fail(CF_ERR_NOMEM, "cf_listing_sizes: out of host memory");
    } catch (const std::exception& e) {
        return fail(CF_ERR_INVALID, std::string("cf_listing_sizes: ") + e.what());
    }
}

extern "C" int cf_listing_names(const cf_listing* l, int64_t lo, int64_t hi, char* out, int64_t capacity, int64_t* bounds, int64_t* needed) {
    if (!listing_range_ok(l, lo, hi)) return fail(CF_ERR_INVALID, "cf_listing_names: bad range");
    int64_t total = 0;
    for (int64_t i = lo; i < hi; ++i) total += (int64_t)strlen(l->blob.data() + l->at[(size_t)i]) + 1;
    if (needed) *needed = total;
    if (!out) return CF_OK;                                  // size query
    if (capacity < total || (!bounds && hi > lo)) return fail(CF_ERR_INVALID, "cf_listing_names: buffer too small");
    int64_t pos = 0;
    for (int64_t i = lo; i < hi; ++i) {
        const char* s = l->blob.data() + l->at[(size_t)i];
        const size_t len = strlen(s) + 1;
        bounds[i - lo] = pos;
        memcpy(out + pos, s, len);
        pos += (int64_t)len;
    }
    if (bounds) bounds[hi - lo] = pos;
    return CF_OK;
}

// A listing from names somebody else read and ordered (rank 0 of a sharded job reads the directory ONCE and broadcasts the ordered
// names: on overlayfs -- the GPU boxes' /tmp -- concurrent readdirs of one directory serialise, 8 ranks x 100 000 entries took 102 ms
// each against 13 ms for one reader; tools/exp_listing.py).  names: n_entries NUL-terminated strings back to back, in order.
extern "C" int cf_listing_from_names(const char* dir, const char* names, int64_t n_bytes, int64_t n_entries, cf_listing** out, uint64_t* digest) {
    if (!dir || !out || n_bytes < 0 || n_entries < 0 || (!names && n_bytes > 0)) return fail(CF_ERR_INVALID, "cf_listing_from_names: bad argument");
    *out = nullptr;
    if (n_bytes > 0xffffffffll) return fail(CF_ERR_INVALID, "cf_listing_from_names: more than 4 GiB of names");
    if (n_bytes > 0 && names[n_bytes - 1] != 0) return fail(CF_ERR_INVALID, "cf_listing_from_names: the last name is not terminated");
    try {
        std::unique_ptr<cf_listing> l(new cf_listing);
        l->dir = dir;
        l->blob.assign(names, names + n_bytes);
        l->at.reserve((size_t)n_entries);
        uint64_t h0 = 1469598103934665603ull, h1 = 0x9e3779b97f4a7c15ull;
        int64_t pos = 0;
        while (pos < n_bytes) {
            if ((int64_t)l->at.size() == n_entries) return fail(CF_ERR_INVALID, "cf_listing_from_names: more names than n_entries");
            l->at.push_back((uint32_t)pos);
            const char* s = l->blob.data() + pos;
            const size_t len = strlen(s);
            if (len == 0) return fail(CF_ERR_INVALID, "cf_listing_from_names: empty name");
            if (!l->at.empty() && l->at.size() > 1 && strcmp(l->blob.data() + l->at[l->at.size() - 2], s) >= 0)
                return fail(CF_ERR_INVALID, "cf_listing_from_names: names are not in strictly ascending bytewise order");
            for (const char* c = s;; ++c) {
                h0 = (h0 ^ (unsigned char)*c) * 1099511628211ull;
                h1 = (h1 ^ (unsigned char)*c) * 0x100000001b3ull + 0x632be59bd9b4e019ull;
                if (!*c) break;
            }
            pos += (int64_t)len + 1;
        }
        if ((int64_t)l->at.size() != n_entries) return fail(CF_ERR_INVALID, "cf_listing_from_names: fewer names than n_entries");
        if (digest) { digest[0] = h0; digest[1] = h1; }
        *out = l.release();
        return CF_OK;
    } catch (const std::bad_alloc&) {
        return fail(CF_ERR_NOMEM, "cf_listing_from_names: out of host memory");
    } catch (const std::exception& e) {
        return fail(CF_ERR_INVALID, std::string("cf_listing_from_names: ") + e.what());
    }
}

// cf_load_npy_int16 for entries [lo, hi) of a listing: the names never leave the library (openat relative to the listing's directory),
// so a rank of the CLI builds no path string per file at all.  Every entry must be named *.npy (infer.load_dac's own dispatch:
// anything else is for the general loader) -- CF_ERR_INVALID names the first that is not, or that is not such an array.
extern "C" int cf_listing_load_npy_int16(const cf_listing* l, int64_t lo, int64_t hi, int16_t* out, int64_t capacity, int64_t* lengths,
                                         int64_t* total, int32_t n_threads) {
    if (!listing_range_ok(l, lo, hi)) return fail(CF_ERR_INVALID, "cf_listing_load_npy_int16: bad range");
    if (total) *total = 0;
    if (hi == lo) return CF_OK;
    int dfd = -1;
    try {
        const int64_t n = hi - lo;
        std::vector<int64_t> bounds((size_t)n + 1, 0);
        for (int64_t i = 0; i < n; ++i) {
            bounds[(size_t)i] = (int64_t)l->at[(size_t)(lo + i)];
            const char* nm = l->blob.data() + bounds[(size_t)i];
            const size_t len = strlen(nm);
            if (len < 4 || memcmp(nm + len - 4, ".npy", 4) != 0)
                return fail(CF_ERR_INVALID, std::string("cf_listing_load_npy_int16: not a .npy file: ") + nm);
        }
        dfd = open(l->dir.c_str(), O_RDONLY | O_DIRECTORY | O_CLOEXEC);
        if (dfd < 0) return fail(CF_ERR_INVALID, std::string("cf_listing_load_npy_int16: cannot open directory ") + l->dir + ": " + strerror(errno));
        const int rc = load_npy_int16(dfd, l->blob.data(), bounds.data(), n, out, capacity, lengths, total, n_threads);
        close(dfd);
        return rc;
    } catch (const std::bad_alloc&) {
        if (dfd >= 0) close(dfd);
        return fail(CF_ERR_NOMEM, "cf_listing_load_npy_int16: out of host memory");
    } catch (const std::exception& e) {
        if (dfd >= 0) close(dfd);
        return fail(CF_ERR_INVALID, std::string("cf_listing_load_npy_int16: ") + e.what());
    }
}

// CRC-32C (Castagnoli, reflected 0x82F63B78) as leveldb tables and TensorFlow checkpoint-V2 bundles use it (every block of
// ckpnt-N.index and every tensor of ckpnt-N.data carries one; catfish/models/rnn_class.py:191-198 restores through tf.train.Saver, which
// verifies them).  Slicing-by-8 over tables built once; catfish_amd/checkpoint.py calls it instead of a byte loop in Python (0.79 MB of
// inference tensors: 45 ms -> 0.4 ms per model load).
extern "C" uint32_t cf_crc32c(const void* data, int64_t n, uint32_t crc) {
    static uint32_t T[8][256];
    static const bool ready = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 255u];
        return true;
    }();
    (void)ready;
    if (!data || n <= 0) return crc;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7u)) { c = T[0][(c ^ *p++) & 255u] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= c;                                              // little-endian: the low four bytes meet the running CRC
        c = T[7][w & 255u] ^ T[6][(w >> 8) & 255u] ^ T[5][(w >> 16) & 255u] ^ T[4][(w >> 24) & 255u] ^ T[3][(w >> 32) & 255u] ^
            T[2][(w >> 40) & 255u] ^ T[1][(w >> 48) & 255u] ^ T[0][(w >> 56) & 255u];
        p += 8;
        n -= 8;
    }
    while (n-- > 0) c = T[0][(c ^ *p++) & 255u] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

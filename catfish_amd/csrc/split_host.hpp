// Host side of the split step (no device work): catfish/catfish:85-92 -> split_f5.split_signal (catfish/split_f5.py:8-81) for the read
// format this image can hold.  The reference cuts the raw signal of every read that has homopolymer chunks at the chunk coordinates
// and writes each piece as a file of its own: `<HP dir>/<stem>_<k>` for the merged HP chunks, `<nonHP dir>/<stem>_<k>` for the stretches
// between them, <stem> = the file name up to its FIRST dot (:39,65), k one running index over both loops (:34,57,81).  Its container is
// a gzip-9 HDF5 copy of the input; here a piece is the same int16 samples as a one-dimensional little-endian .npy, byte for byte what
// numpy.save writes (catfish_amd/split.py is the same step in Python, for every format infer.load_dac reads, and the definition the
// tests compare this with).  A rank of the CLI writes ~5 files per read: per file Python costs ~30 us, which made the split ten times
// the classification of the same reads; here a pool of host threads reads each input once (cf_loader::slurp) and writes its pieces
// with one write() each.  Included by catfish_hip.hip after loader_host.hpp.
#include <array>
#include <atomic>

namespace cf_split {

// numpy.lib.format's version-1.0 header of an int16 vector of n samples: magic, u16 length, the dict, blanks up to a multiple of 64, '\n'
static size_t npy_header(int64_t n, char* out /* >= 192 bytes */) {
    char dict[128];
    const int len = snprintf(dict, sizeof dict, "{'descr': '<i2', 'fortran_order': False, 'shape': (%lld,), }", (long long)n);
    const size_t pad = (64 - (10 + (size_t)len + 1) % 64) % 64;
    const size_t hlen = (size_t)len + pad + 1;
    memcpy(out, "\x93NUMPY\x01\x00", 8);
    out[8] = (char)(hlen & 255u);
    out[9] = (char)(hlen >> 8);
    memcpy(out + 10, dict, (size_t)len);
    memset(out + 10 + len, ' ', pad);
    out[10 + hlen - 1] = '\n';
    return 10 + hlen;
}

// Python's slice(s0, s1).indices(n) for step 1: a bound below zero counts from the end, then both are clamped to [0, n]
static void slice_bounds(int64_t s0, int64_t s1, int64_t n, int64_t& a, int64_t& b) {
    if (s0 < 0) s0 += n;
    if (s1 < 0) s1 += n;
    a = std::min(std::max<int64_t>(s0, 0), n);
    b = std::min(std::max<int64_t>(s1, 0), n);
    if (b < a) b = a;
}

static bool write_all(int fd, const unsigned char* p, size_t n) {
    while (n > 0) {
        const ssize_t w = write(fd, p, n);
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) { if (w == 0) errno = EIO; return false; }
        p += w;
        n -= (size_t)w;
    }
    return true;
}

struct Failure { int64_t read = -1; int kind = 0; std::string what; };      // kind 1: not such a read (CF_ERR_INVALID), 2: I/O (CF_ERR_IO)

}  // namespace cf_split

static int listing_split(const cf_listing* l, int64_t lo, int64_t hi, const int64_t* hp_bounds, const int64_t* hp_start, const int64_t* hp_end,
                         const int64_t* non_bounds, const int64_t* non_start, const int64_t* non_end, const char* hp_dir, const char* non_dir,
                         int32_t n_threads, int64_t* counts) {
    if (!listing_range_ok(l, lo, hi)) return fail(CF_ERR_INVALID, "cf_listing_split_npy_int16: bad range");
    if (counts) counts[0] = counts[1] = counts[2] = counts[3] = 0;
    const int64_t n = hi - lo;
    if (n == 0) return CF_OK;
    if (!hp_bounds || !non_bounds || !hp_dir || !non_dir) return fail(CF_ERR_INVALID, "cf_listing_split_npy_int16: null argument");
    if (hp_bounds[n] > hp_bounds[0] && (!hp_start || !hp_end)) return fail(CF_ERR_INVALID, "cf_listing_split_npy_int16: null chunk table");
    if (non_bounds[n] > non_bounds[0] && (!non_start || !non_end)) return fail(CF_ERR_INVALID, "cf_listing_split_npy_int16: null chunk table");
    std::vector<int64_t> todo;                              // rows with homopolymer chunks: `for read in hp_dict` (catfish/catfish:88)
    for (int64_t r = 0; r < n; ++r) {
        if (hp_bounds[r + 1] < hp_bounds[r] || non_bounds[r + 1] < non_bounds[r])
            return fail(CF_ERR_INVALID, "cf_listing_split_npy_int16: descending bounds");
        if (hp_bounds[r + 1] > hp_bounds[r]) todo.push_back(r);
    }
    if (todo.empty()) return CF_OK;
    for (int64_t r : todo) {
        const char* nm = l->blob.data() + l->at[(size_t)(lo + r)];
        const size_t len = strlen(nm);
        if (len < 4 || memcmp(nm + len - 4, ".npy", 4) != 0)
            return fail(CF_ERR_INVALID, std::string("cf_listing_split_npy_int16: not a .npy file: ") + nm);
    }
    // The unit of work is a run of reads that share a stem (`a.b.npy`, `a.c.npy`: neighbours in the listing's bytewise order, both
    // "a"): their pieces have the same names and the later read overwrites the earlier one's, as in the reference's sequential loop --
    // one thread takes them in order.  (Across the blocks of two ranks that order is not kept.)
    auto stem_len = [&](int64_t r) { const char* nm = l->blob.data() + l->at[(size_t)(lo + r)]; const char* d = strchr(nm, '.'); return d ? (size_t)(d - nm) : strlen(nm); };
    std::vector<size_t> group_at(1, 0);                     // todo[group_at[g] .. group_at[g + 1]) share a stem
    for (size_t k = 1; k < todo.size(); ++k) {
        const size_t la = stem_len(todo[k - 1]), lb = stem_len(todo[k]);
        if (la != lb || memcmp(l->blob.data() + l->at[(size_t)(lo + todo[k - 1])], l->blob.data() + l->at[(size_t)(lo + todo[k])], la) != 0)
            group_at.push_back(k);
    }
    group_at.push_back(todo.size());
    const int64_t n_groups = (int64_t)group_at.size() - 1;
    int open_errno = 0;
    const char* unopened = nullptr;
    auto open_dir = [&](const char* path) {
        const int fd = open(path, O_RDONLY | O_DIRECTORY | O_CLOEXEC);
        if (fd < 0 && !unopened) { open_errno = errno; unopened = path; }
        return fd;
    };
    const int in_fd = open_dir(l->dir.c_str());
    const int hp_fd = open_dir(hp_dir);
    const int non_fd = open_dir(non_dir);
    auto close_all = [&] { if (in_fd >= 0) close(in_fd); if (hp_fd >= 0) close(hp_fd); if (non_fd >= 0) close(non_fd); };
    if (unopened) {
        const std::string msg = std::string("cf_listing_split_npy_int16: cannot open ") + unopened + ": " + strerror(open_errno);
        close_all();
        return fail(CF_ERR_IO, msg);
    }
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : 4, std::min<int64_t>(n_groups, 64)));
    std::vector<cf_split::Failure> failed((size_t)nt);
    std::vector<std::array<int64_t, 4>> done((size_t)nt, std::array<int64_t, 4>{0, 0, 0, 0});
    std::atomic<int64_t> next(0);
    std::atomic<bool> stop(false);
    try {
        cf_loader::run_pool(nt, [&](int t) {
            std::vector<unsigned char> out;
            cf_split::Failure& bad = failed[(size_t)t];
            try {
                for (;;) {
                    const int64_t g = next.fetch_add(1);
                    if (g >= n_groups || stop.load()) break;
                    for (size_t k = group_at[(size_t)g]; k < group_at[(size_t)g + 1] && bad.read < 0; ++k) {
                        const int64_t r = todo[k];
                        const char* nm = l->blob.data() + l->at[(size_t)(lo + r)];
                        cf_loader::Item it;
                        if (!cf_loader::slurp(in_fd, nm, it)) {
                            bad.read = r; bad.kind = 1;
                            bad.what = std::string("not a readable one-dimensional little-endian int16 .npy: ") + nm;
                            stop.store(true);
                            break;
                        }
                        const int16_t* sig = reinterpret_cast<const int16_t*>(it.bytes.data() + it.data_off);
                        const char* dot = strchr(nm, '.');
                        const std::string stem(nm, dot ? (size_t)(dot - nm) : strlen(nm));
                        int64_t index = 0;
                        for (int part = 0; part < 2 && bad.read < 0; ++part) {
                            const int64_t* bounds = part ? non_bounds : hp_bounds;
                            const int64_t* st = part ? non_start : hp_start;
                            const int64_t* en = part ? non_end : hp_end;
                            for (int64_t row = bounds[r]; row < bounds[r + 1]; ++row, ++index) {
                                int64_t a, b;
                                cf_split::slice_bounds(st[row], en[row], it.count, a, b);
                                char head[192];
                                const size_t hl = cf_split::npy_header(b - a, head);
                                out.resize(hl + 2 * (size_t)(b - a));
                                memcpy(out.data(), head, hl);
                                if (b > a) memcpy(out.data() + hl, sig + a, 2 * (size_t)(b - a));
                                const std::string dest = stem + "_" + std::to_string(index) + ".npy";
                                const int fd = openat(part ? non_fd : hp_fd, dest.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
                                bool ok = fd >= 0;
                                int err = ok ? 0 : errno;
                                if (ok && !cf_split::write_all(fd, out.data(), out.size())) { ok = false; err = errno; }
                                if (fd >= 0 && close(fd) != 0 && ok) { ok = false; err = errno; }
                                if (!ok) {
                                    bad.read = r; bad.kind = 2;
                                    bad.what = std::string(part ? non_dir : hp_dir) + "/" + dest + ": " + strerror(err);
                                    stop.store(true);
                                    break;
                                }
                                done[(size_t)t][1 + part] += 1;
                                done[(size_t)t][3] += b - a;
                            }
                        }
                        if (bad.read < 0) done[(size_t)t][0] += 1;
                    }
                }
            } catch (const std::bad_alloc&) {                   // a thread may not throw
                bad.read = 0; bad.kind = 3; bad.what = "out of host memory";
                stop.store(true);
            } catch (...) {
                bad.read = 0; bad.kind = 2; bad.what = "unexpected exception in a split thread";
                stop.store(true);
            }
        });
    } catch (...) {
        close_all();
        throw;
    }
    close_all();
    const cf_split::Failure* first = nullptr;
    for (const cf_split::Failure& f : failed)
        if (f.read >= 0 && (!first || f.read < first->read)) first = &f;
    if (first)
        return fail(first->kind == 1 ? CF_ERR_INVALID : first->kind == 3 ? CF_ERR_NOMEM : CF_ERR_IO, "cf_listing_split_npy_int16: " + first->what);
    if (counts)
        for (const auto& d : done)
            for (int i = 0; i < 4; ++i) counts[i] += d[(size_t)i];
    return CF_OK;
}

extern "C" int cf_listing_split_npy_int16(const cf_listing* l, int64_t lo, int64_t hi, const int64_t* hp_bounds, const int64_t* hp_start,
                                          const int64_t* hp_end, const int64_t* nonhp_bounds, const int64_t* nonhp_start, const int64_t* nonhp_end,
                                          const char* hp_dir, const char* nonhp_dir, int32_t n_threads, int64_t* counts) {
    try {
        return listing_split(l, lo, hi, hp_bounds, hp_start, hp_end, nonhp_bounds, nonhp_start, nonhp_end, hp_dir, nonhp_dir, n_threads, counts);
    } catch (const std::bad_alloc&) {
        return fail(CF_ERR_NOMEM, "cf_listing_split_npy_int16: out of host memory");
    } catch (const std::exception& e) {
        return fail(CF_ERR_INVALID, std::string("cf_listing_split_npy_int16: ") + e.what());
    }
}

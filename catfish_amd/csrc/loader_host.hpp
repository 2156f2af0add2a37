// Host side of the ingest (no device work): the file read of the reference's per-file loop (catfish/catfish:50-56 ->
// infer.process_signal, infer.py:77-93) for the read format this image can hold -- there is no HDF5 library here, so a read
// is a one-dimensional C-order little-endian int16 .npy (DAC codes after the leader trim).  Many files are read by a small
// pool of host threads straight into ONE caller-owned buffer (the pinned staging buffer of the streaming pipeline), back
// to back in the order given: the per-file Python loader costs ~15 us a file under the GIL, which is what bounded a rank of
// the CLI (12 500 x 4096-sample files: 0.19 s of loading against 0.16 s of device work).  Included by catfish_hip.hip.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <thread>

namespace cf_loader {

struct Item {
    std::vector<unsigned char> bytes;    // whole file
    size_t data_off = 0;
    int64_t count = -1;                  // samples; < 0: not a plain int16 vector (or unreadable)
};

// the byte-level header check of catfish_amd/infer.py::_read_npy_int16: magic, version 1 or 2, '<i2', C order, one dimension,
// and the file exactly as long as header + data
static bool parse(Item& it) {
    const std::vector<unsigned char>& b = it.bytes;
    if (b.size() < 12 || memcmp(b.data(), "\x93NUMPY", 6) != 0 || (b[6] != 1 && b[6] != 2)) return false;
    size_t hlen, off;
    if (b[6] == 1) { hlen = (size_t)b[8] | ((size_t)b[9] << 8); off = 10; }
    else { hlen = (size_t)b[8] | ((size_t)b[9] << 8) | ((size_t)b[10] << 16) | ((size_t)b[11] << 24); off = 12; }
    if (off + hlen > b.size()) return false;
    const std::string header(reinterpret_cast<const char*>(b.data() + off), hlen);
    if (header.find("'descr': '<i2'") == std::string::npos || header.find("'fortran_order': False") == std::string::npos) return false;
    const size_t a = header.find("'shape': (");
    if (a == std::string::npos) return false;
    const size_t close = header.find(')', a);
    if (close == std::string::npos) return false;
    // exactly one dimension: digits, optional blanks, one comma
    size_t p = a + 10;
    while (p < close && header[p] == ' ') ++p;
    int64_t n = 0;
    size_t digits = 0;
    while (p < close && header[p] >= '0' && header[p] <= '9') { n = n * 10 + (header[p] - '0'); ++p; ++digits; if (digits > 15) return false; }
    if (digits == 0) return false;
    int commas = 0;
    for (; p < close; ++p) {
        if (header[p] == ',') ++commas;
        else if (header[p] != ' ') return false;
    }
    if (commas > 1) return false;
    if (b.size() != off + hlen + 2 * (size_t)n) return false;
    it.data_off = off + hlen;
    it.count = n;
    return true;
}

static bool slurp(const char* path, Item& it) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 0) { close(fd); return false; }
    it.bytes.resize((size_t)st.st_size);
    size_t got = 0;
    while (got < it.bytes.size()) {
        const ssize_t r = pread(fd, it.bytes.data() + got, it.bytes.size() - got, (off_t)got);
        if (r <= 0) break;
        got += (size_t)r;
    }
    close(fd);
    return got == it.bytes.size();
}

}  // namespace cf_loader

extern "C" int cf_load_npy_int16(const char* paths, const int64_t* path_bounds, int64_t n_files, int16_t* out, int64_t capacity,
                                 int64_t* lengths, int64_t* total, int32_t n_threads) {
    if (n_files < 0 || capacity < 0) return fail(CF_ERR_INVALID, "cf_load_npy_int16: negative size");
    if (total) *total = 0;
    if (n_files == 0) return CF_OK;
    if (!paths || !path_bounds || !lengths || (!out && capacity > 0)) return fail(CF_ERR_INVALID, "cf_load_npy_int16: null buffer");
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : 4, std::min<int64_t>(n_files, 64)));
    std::vector<cf_loader::Item> items((size_t)n_files);
    std::vector<int64_t> bad((size_t)nt, -1);                    // first offending file per thread
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t)
            pool.emplace_back([&, t]() {
                for (int64_t i = t; i < n_files; i += nt) {      // interleaved: neighbouring files are neighbours on disk
                    cf_loader::Item& it = items[(size_t)i];
                    if (!cf_loader::slurp(paths + path_bounds[i], it) || !cf_loader::parse(it)) {
                        it.count = -1;
                        if (bad[(size_t)t] < 0) bad[(size_t)t] = i;
                    }
                }
            });
        for (std::thread& th : pool) th.join();
    }
    int64_t first_bad = -1;
    for (int64_t b : bad)
        if (b >= 0 && (first_bad < 0 || b < first_bad)) first_bad = b;
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
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t)
            pool.emplace_back([&, t]() {
                for (int64_t i = t; i < n_files; i += nt) {
                    const cf_loader::Item& it = items[(size_t)i];
                    memcpy(out + offs[(size_t)i], it.bytes.data() + it.data_off, 2 * (size_t)it.count);
                }
            });
        for (std::thread& th : pool) th.join();
    }
    return CF_OK;
}

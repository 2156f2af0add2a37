// ThreadSanitizer driver for the host-only entry points that run thread pools (catfish_amd/csrc/loader_host.hpp): cf_load_npy_int16,
// cf_stat_files, cf_listing_sizes, cf_listing_load_npy_int16 -- and independent listings used from two caller threads at once, which
// include/catfish_hip.h declares safe.  Built and run by tests/test_host_sanitizers.py:
//   g++ -std=c++17 -O1 -g -fsanitize=thread -o tsan_driver tests/native/tsan_driver.cpp -lpthread && ./tsan_driver <scratch dir>
// Exit code 0 and "tsan ok" = every call returned what it should; a data race makes TSan print a report (the test looks for it).
#include "host_entry_shim.cpp"

#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <thread>

static void write_npy(const std::string& path, const std::vector<int16_t>& v) {
    std::string head = "{'descr': '<i2', 'fortran_order': False, 'shape': (" + std::to_string(v.size()) + ",), }";
    while ((10 + head.size() + 1) % 64 != 0) head += ' ';
    head += '\n';
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    const unsigned char magic[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(head.size() & 255), (unsigned char)(head.size() >> 8)};
    fwrite(magic, 1, 10, f);
    fwrite(head.data(), 1, head.size(), f);
    fwrite(v.data(), 2, v.size(), f);
    fclose(f);
}

#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, cf_last_error()); exit(1); } \
    } while (0)

static void exercise(const std::string& dir, int n_files, int rounds) {
    cf_listing* l = nullptr;
    int64_t n = -1;
    uint64_t dig[2];
    CHECK(cf_listing_open(dir.c_str(), &l, &n, dig) == CF_OK && n == n_files);
    std::vector<int64_t> sizes((size_t)n), lengths((size_t)n);
    int64_t expect_total = 0;
    for (int r = 0; r < rounds; ++r) {
        const int threads = 1 + (r % 8);
        CHECK(cf_listing_sizes(l, 0, n, sizes.data(), threads) == CF_OK);
        int64_t total = -1;
        int64_t cap = 0;
        for (int64_t i = 0; i < n; ++i) cap += (sizes[(size_t)i] - 64) / 2 + 64;
        std::vector<int16_t> out((size_t)cap);
        CHECK(cf_listing_load_npy_int16(l, 0, n, out.data(), cap, lengths.data(), &total, threads) == CF_OK);
        if (r == 0) expect_total = total;
        CHECK(total == expect_total && total > 0);
        int64_t pos = 0;                                          // every file holds i, i+1, ... in its samples' low bits: check the copy order
        for (int64_t i = 0; i < n; ++i) {
            for (int64_t k = 0; k < lengths[(size_t)i]; ++k) CHECK(out[(size_t)(pos + k)] == (int16_t)((lengths[(size_t)i] + k) & 0x7fff));
            pos += lengths[(size_t)i];
        }
        // the same block through the path-based entry point and the plain stat call
        int64_t need = 0;
        CHECK(cf_listing_names(l, 0, n, nullptr, 0, nullptr, &need) == CF_OK);
        std::vector<char> names((size_t)need);
        std::vector<int64_t> nb((size_t)n + 1);
        CHECK(cf_listing_names(l, 0, n, names.data(), need, nb.data(), nullptr) == CF_OK);
        std::vector<int64_t> sizes2((size_t)n);
        CHECK(cf_stat_files(dir.c_str(), names.data(), nb.data(), n, sizes2.data(), threads) == CF_OK && sizes2 == sizes);
        std::string paths;
        std::vector<int64_t> pb((size_t)n + 1);
        for (int64_t i = 0; i < n; ++i) {
            pb[(size_t)i] = (int64_t)paths.size();
            paths += dir + "/" + (names.data() + nb[(size_t)i]);
            paths.push_back('\0');
        }
        pb[(size_t)n] = (int64_t)paths.size();
        std::vector<int16_t> out2((size_t)cap);
        int64_t total2 = -1;
        CHECK(cf_load_npy_int16(paths.data(), pb.data(), n, out2.data(), cap, lengths.data(), &total2, threads) == CF_OK && total2 == total);
        CHECK(memcmp(out.data(), out2.data(), (size_t)total * 2) == 0);
        // the split step over the same listing: two HP rows and one non-HP row per read (every third read has none and is skipped),
        // pieces written by the pool; counts add up and a piece holds the slice
        if (r < 4) {
            const std::string hp = dir + "_HP", non = dir + "_nonHP";      // beside the input directory: its listing stays as it is
            mkdir(hp.c_str(), 0755);
            mkdir(non.c_str(), 0755);
            std::vector<int64_t> hb((size_t)n + 1, 0), nbnd((size_t)n + 1, 0), hs, he, ns, ne;
            int64_t want_reads = 0, want_samples = 0;
            for (int64_t i = 0; i < n; ++i) {
                const int64_t len = lengths[(size_t)i];
                if (i % 3 != 2) {
                    hs.push_back(0); he.push_back(len / 2); hs.push_back(-5); he.push_back(len + 9);
                    ns.push_back(len / 2); ne.push_back(len);
                    ++want_reads;
                    want_samples += len / 2 + std::min<int64_t>(5, len) + (len - len / 2);
                }
                hb[(size_t)i + 1] = (int64_t)hs.size();
                nbnd[(size_t)i + 1] = (int64_t)ns.size();
            }
            int64_t counts[4] = {-1, -1, -1, -1};
            CHECK(cf_listing_split_npy_int16(l, 0, n, hb.data(), hs.data(), he.data(), nbnd.data(), ns.data(), ne.data(), hp.c_str(), non.c_str(),
                                             threads, counts) == CF_OK);
            CHECK(counts[0] == want_reads && counts[1] == 2 * want_reads && counts[2] == want_reads && counts[3] == want_samples);
            cf_loader::Item piece;
            const std::string first = std::string(names.data() + nb[0]);
            const std::string rel = first.substr(0, first.find('.')) + "_0.npy";
            const int dfd = open(hp.c_str(), O_RDONLY | O_DIRECTORY);
            CHECK(dfd >= 0 && cf_loader::slurp(dfd, rel.c_str(), piece) && piece.count == lengths[0] / 2);
            close(dfd);
        }
    }
    cf_listing_close(l);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: tsan_driver <scratch dir>\n"); return 2; }
    const std::string root = argv[1];
    const int n_files = 240;
    for (int d = 0; d < 2; ++d) {
        const std::string dir = root + "/reads" + std::to_string(d);
        mkdir(dir.c_str(), 0755);
        for (int i = 0; i < n_files; ++i) {
            const size_t len = (size_t)(1 + (i * 37 + d * 11) % 3000);
            std::vector<int16_t> v(len);
            for (size_t k = 0; k < len; ++k) v[k] = (int16_t)((len + k) & 0x7fff);
            char name[64];
            snprintf(name, sizeof name, "read_%04d.npy", (i * 7919) % 10000);
            write_npy(dir + "/" + name, v);
        }
    }
    // one listing per caller thread, both at once (independent listings are safe to use concurrently)
    std::thread a(exercise, root + "/reads0", n_files, 12);
    std::thread b(exercise, root + "/reads1", n_files, 12);
    a.join();
    b.join();
    // the error path from several pool threads at once: three files that are not such arrays among good ones
    const std::string bad = root + "/reads0";
    FILE* f = fopen((bad + "/aaa_broken.npy").c_str(), "wb"); fputs("junk", f); fclose(f);
    f = fopen((bad + "/mmm_broken.npy").c_str(), "wb"); fputs("\x93NUMPY\x01", f); fclose(f);
    f = fopen((bad + "/zzz_broken.npy").c_str(), "wb"); fclose(f);
    cf_listing* l = nullptr;
    int64_t n = 0;
    CHECK(cf_listing_open(bad.c_str(), &l, &n, nullptr) == CF_OK && n == n_files + 3);
    std::vector<int16_t> out(1 << 20);
    std::vector<int64_t> lengths((size_t)n);
    int64_t total = 0;
    for (int t = 1; t <= 8; ++t) {
        CHECK(cf_listing_load_npy_int16(l, 0, n, out.data(), (int64_t)out.size(), lengths.data(), &total, t) == CF_ERR_INVALID);
        CHECK(strstr(cf_last_error(), "aaa_broken.npy") != nullptr);              // the FIRST offending file, whichever thread met it
    }
    cf_listing_close(l);
    printf("tsan ok\n");
    return 0;
}

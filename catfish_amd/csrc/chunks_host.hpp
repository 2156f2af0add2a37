// Host side of the pipeline tail (no device work): what the reference's per-file loop does with a read's homopolymer
// spans after infer_class_from_signal returns them (catfish/catfish:57-82, center_hp :121-135) -- greedy merge into
// chunks of at least chunk_size samples, each chunk widened around its centre and pushed back inside the read, then
// the complement -- over MANY reads held as flat arrays, and the JSON text of the result.  Included by
// catfish_hip.hip; every rank of a sharded job runs it on its own reads.
//
// The reference mutates its span lists in place and its merged list holds ALIASES of them, which is observable: when
// the first span alone is >= chunk_size long the same list object ends up in the merged list twice and later edits
// show in both places, and `hp_positions[i - 1]` at i = 0 is the LAST span.  So the merged list is kept here as
// indices into a working copy of the read's spans, and values are read out only at the end.

namespace cf_chunks {

struct Span { int64_t a, b; };

// center_hp (catfish/catfish:121-135) on one span
static inline void center(Span& s, int64_t len_read, int64_t chunk) {
    const int64_t len_hp = s.b - s.a;
    if (len_hp >= chunk) return;
    const int64_t left = (chunk - len_hp) / 2;            // chunk - len_hp > 0: floor == truncation
    const int64_t right = (chunk - len_hp) - left;
    s.a -= left;
    s.b += right;
    if (s.a < 0) { s.b -= s.a; s.a = 0; }
    if (s.b > len_read) { s.a -= len_read - s.b; s.b = len_read; }
}

}  // namespace cf_chunks

static int chunks_from_spans(const int64_t* span_bounds, const int64_t* span_start, const int64_t* span_end,
                             const int64_t* lengths, int64_t n_reads, int64_t chunk_size,
                             int64_t* hp_bounds, int64_t* hp_start, int64_t* hp_end, int64_t hp_capacity,
                             int64_t* nonhp_bounds, int64_t* nonhp_start, int64_t* nonhp_end, int64_t nonhp_capacity) {
    using cf_chunks::Span;
    if (n_reads < 0 || hp_capacity < 0 || nonhp_capacity < 0) return fail(CF_ERR_INVALID, "cf_chunks_from_spans: negative size");
    if (!span_bounds || !lengths || !hp_bounds || !nonhp_bounds) return fail(CF_ERR_INVALID, "cf_chunks_from_spans: null table");
    if ((hp_capacity > 0 && (!hp_start || !hp_end)) || (nonhp_capacity > 0 && (!nonhp_start || !nonhp_end)))
        return fail(CF_ERR_INVALID, "cf_chunks_from_spans: null output table");
    const int64_t n_spans = span_bounds[n_reads] - span_bounds[0];
    if (n_spans < 0 || (n_spans > 0 && (!span_start || !span_end))) return fail(CF_ERR_INVALID, "cf_chunks_from_spans: bad span table");
    std::vector<Span> work;
    std::vector<int64_t> merged;
    int64_t n_hp = 0, n_non = 0;
    hp_bounds[0] = nonhp_bounds[0] = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        const int64_t lo = span_bounds[r], n = span_bounds[r + 1] - lo, len_read = lengths[r];
        if (n < 0) return fail(CF_ERR_INVALID, "cf_chunks_from_spans: span_bounds must ascend");
        if (n == 0) {                                      // catfish:82: one stretch covering the read (its odd nesting is the formatter's business)
            if (n_non + 1 > nonhp_capacity) return fail(CF_ERR_INVALID, "cf_chunks_from_spans: nonhp_capacity too small");
            nonhp_start[n_non] = 0;
            nonhp_end[n_non++] = len_read;
        } else {
            work.resize((size_t)n);
            for (int64_t i = 0; i < n; ++i) work[(size_t)i] = Span{span_start[lo + i], span_end[lo + i]};
            merged.assign(1, 0);
            for (int64_t i = 0; i < n; ++i) {
                if (work[(size_t)i].b >= chunk_size + work[(size_t)merged.back()].a) {
                    work[(size_t)merged.back()].b = work[(size_t)(i == 0 ? n - 1 : i - 1)].b;
                    cf_chunks::center(work[(size_t)merged.back()], len_read, chunk_size);
                    merged.push_back(i);
                }
            }
            cf_chunks::center(work[(size_t)merged.back()], len_read, chunk_size);
            if (n_hp + (int64_t)merged.size() > hp_capacity) return fail(CF_ERR_INVALID, "cf_chunks_from_spans: hp_capacity too small");
            if (n_non + (int64_t)merged.size() + 1 > nonhp_capacity) return fail(CF_ERR_INVALID, "cf_chunks_from_spans: nonhp_capacity too small");
            int64_t m_start = 0;
            for (int64_t k : merged) {
                const Span s = work[(size_t)k];
                hp_start[n_hp] = s.a;
                hp_end[n_hp++] = s.b;
                if (s.a > m_start) { nonhp_start[n_non] = m_start; nonhp_end[n_non++] = s.a - 1; }
                m_start = s.b;
            }
            if (m_start != len_read) { nonhp_start[n_non] = m_start; nonhp_end[n_non++] = len_read; }
        }
        hp_bounds[r + 1] = n_hp;
        nonhp_bounds[r + 1] = n_non;
    }
    return CF_OK;
}

extern "C" int cf_chunks_from_spans(const int64_t* span_bounds, const int64_t* span_start, const int64_t* span_end,
                                    const int64_t* lengths, int64_t n_reads, int64_t chunk_size,
                                    int64_t* hp_bounds, int64_t* hp_start, int64_t* hp_end, int64_t hp_capacity,
                                    int64_t* nonhp_bounds, int64_t* nonhp_start, int64_t* nonhp_end, int64_t nonhp_capacity) {
    try {       // (the working copies are std::vectors: no bad_alloc across the C ABI)
        return chunks_from_spans(span_bounds, span_start, span_end, lengths, n_reads, chunk_size, hp_bounds, hp_start, hp_end, hp_capacity,
                                 nonhp_bounds, nonhp_start, nonhp_end, nonhp_capacity);
    } catch (const std::bad_alloc&) {
        return fail(CF_ERR_NOMEM, "cf_chunks_from_spans: out of host memory");
    }
}

// JSON members `"name": [[a, b], [c, d]]`, joined by ", " (what json.dump writes between the braces of a dict of lists of
// pairs, default separators), for the reads that own at least one row of the table (whole_read given: for every read, `[]`
// when it owns none -- the reference's nonhp_dict has an entry per read, its hp_dict only for reads with homopolymers).
// keys: the reads' names as JSON string literals back to back, key_bounds[n_reads + 1] their byte offsets.  whole_read (may be NULL): reads with whole_read[r] != 0
// are written in the reference's no-homopolymer form `[[[a, b], b]]` (catfish:82) from their single row.  Returns the
// number of bytes written, or the negative error code (CF_ERR_INVALID when capacity is too small).
extern "C" int64_t cf_chunks_json(const char* keys, const int64_t* key_bounds, int64_t n_reads, const int64_t* bounds,
                                  const int64_t* start, const int64_t* end, const uint8_t* whole_read, char* out,
                                  int64_t capacity) {
    if (n_reads < 0 || capacity < 0 || !key_bounds || !bounds || !out || (n_reads > 0 && !keys))
        return fail(CF_ERR_INVALID, "cf_chunks_json: bad arguments");
    int64_t w = 0;
    auto put_int = [&](int64_t v) {
        char tmp[24];
        int k = 0;
        uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
        do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) out[w++] = '-';
        while (k) out[w++] = tmp[--k];
    };
    bool first = true;
    for (int64_t r = 0; r < n_reads; ++r) {
        const int64_t lo = bounds[r], hi = bounds[r + 1];
        if (hi <= lo && !whole_read) continue;
        const int64_t klen = key_bounds[r + 1] - key_bounds[r];
        if (klen < 0 || (hi > lo && (!start || !end))) return fail(CF_ERR_INVALID, "cf_chunks_json: bad tables");
        if (w + klen + 8 + (hi - lo) * 48 + 32 > capacity) return fail(CF_ERR_INVALID, "cf_chunks_json: capacity too small");
        if (!first) { out[w++] = ','; out[w++] = ' '; }
        first = false;
        memcpy(out + w, keys + key_bounds[r], (size_t)klen);
        w += klen;
        out[w++] = ':'; out[w++] = ' '; out[w++] = '[';
        if (whole_read && whole_read[r] && hi > lo) {
            out[w++] = '['; out[w++] = '[';
            put_int(start[lo]); out[w++] = ','; out[w++] = ' '; put_int(end[lo]);
            out[w++] = ']'; out[w++] = ','; out[w++] = ' '; put_int(end[lo]); out[w++] = ']';
        } else {
            for (int64_t i = lo; i < hi; ++i) {
                if (i > lo) { out[w++] = ','; out[w++] = ' '; }
                out[w++] = '['; put_int(start[i]); out[w++] = ','; out[w++] = ' '; put_int(end[i]); out[w++] = ']';
            }
        }
        out[w++] = ']';
    }
    return w;
}

#include "graph.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <fstream>
#include <memory>
#include <mutex>
#include <set>
#include <thread>

#include "io.hpp"
#include "overlap.hpp"
#include "pile.hpp"
#include "rala_hip.h"
#include "sequence.hpp"

namespace rala {

namespace {

struct StageTimer {      // same stage lines as the reference's logger (graph.cpp:246,266,384,...)
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void operator()() { t0 = std::chrono::steady_clock::now(); }
    void operator()(const char* msg) {
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "%s %.6lf s\n", msg, s);
    }
};

void check(rala_hip_ctx* ctx, int rc, const char* where) {
    if (rc == RALA_HIP_OK) return;
    fprintf(stderr, "[rala::Graph::%s] error: %s!\n", where, rala_hip_last_error(ctx));
    exit(1);
}

}  // namespace

std::unique_ptr<Graph> createGraph(const std::string& sequences_path, const std::string& overlaps_path,
    uint32_t num_threads) {
    using io::has_suffix;
    if (!(has_suffix(sequences_path, ".fasta") || has_suffix(sequences_path, ".fa") ||
          has_suffix(sequences_path, ".fasta.gz") || has_suffix(sequences_path, ".fa.gz") ||
          has_suffix(sequences_path, ".fastq") || has_suffix(sequences_path, ".fq") ||
          has_suffix(sequences_path, ".fastq.gz") || has_suffix(sequences_path, ".fq.gz"))) {
        fprintf(stderr, "[rala::createGraph] error: "
            "file %s has unsupported format extension (valid extensions: "
            ".fasta, .fasta.gz, .fa, .fa.gz, .fastq, .fastq.gz, .fq, .fq.gz)!\n", sequences_path.c_str());
        exit(1);
    }
    if (!(has_suffix(overlaps_path, ".mhap") || has_suffix(overlaps_path, ".mhap.gz") ||
          has_suffix(overlaps_path, ".paf") || has_suffix(overlaps_path, ".paf.gz"))) {
        fprintf(stderr, "[rala::createGraph] error: "
            "file %s has unsupported format extension (valid extensions: "
            ".mhap, .mhap.gz, .paf, .paf.gz)!\n", overlaps_path.c_str());
        exit(1);
    }
    return std::unique_ptr<Graph>(new Graph(sequences_path, overlaps_path, num_threads));
}

Graph::Graph(const std::string& sequences_path, const std::string& overlaps_path, uint32_t num_threads)
        : sequences_path_(sequences_path), overlaps_path_(overlaps_path), num_threads_(num_threads), ctx_(nullptr) {
    open_devices();
}

// RALA_GPUS = number of devices (default 1); RALA_GPU_DEVICES = their ordinals, comma separated
// (default 0, 1, ...); RALA_COMM = rccl (default) | local (in-process transport, also several ranks
// on one device).  The reference creates its thread pool here (graph.cpp:230-238).
void Graph::open_devices() {
    uint32_t n_gpus = 1;
    if (const char* e = getenv("RALA_GPUS")) n_gpus = (uint32_t)std::max(1, atoi(e));
    if (n_gpus == 1) {
        if (rala_hip_create(0, &ctx_) != RALA_HIP_OK) {
            fprintf(stderr, "[rala::Graph::Graph] error: no usable HIP device!\n");
            exit(1);
        }
        return;
    }
    if (n_gpus > 64) {
        fprintf(stderr, "[rala::Graph::Graph] error: at most 64 devices!\n");
        exit(1);
    }
    std::vector<int> device(n_gpus);
    for (uint32_t k = 0; k < n_gpus; ++k) device[k] = (int)k;
    if (const char* e = getenv("RALA_GPU_DEVICES")) {
        const char* p = e;
        for (uint32_t k = 0; k < n_gpus && *p; ++k) {
            device[k] = atoi(p);
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
    }
    const char* transport = getenv("RALA_COMM");
    const bool local = transport != nullptr && std::string(transport) == "local";
    unsigned char id[128] = {0};
    if (local) {
        if (rala_hip_mg_local_group_create(n_gpus, &local_group_) != RALA_HIP_OK) {
            fprintf(stderr, "[rala::Graph::Graph] error: cannot set up %u ranks!\n", n_gpus);
            exit(1);
        }
    } else if (rala_hip_mg_unique_id(id) != RALA_HIP_OK) {
        fprintf(stderr, "[rala::Graph::Graph] error: RCCL is not usable (RALA_COMM=local selects the in-process transport)!\n");
        exit(1);
    }
    // every rank's device contexts first (not collective): nobody enters ncclCommInitRank unless all ranks have theirs -
    // a rank that failed before it would leave the others waiting inside
    ranks_.assign(n_gpus, nullptr);
    std::vector<int> rc(n_gpus, 0);
    {
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < n_gpus; ++k) {
            th.emplace_back([&, k]() { rc[k] = rala_hip_mg_create_contexts(device[k], k, n_gpus, &ranks_[k]); });
        }
        for (auto& t : th) t.join();
    }
    for (uint32_t k = 0; k < n_gpus; ++k) {
        if (rc[k] != RALA_HIP_OK) {
            fprintf(stderr, "[rala::Graph::Graph] error: device %d (rank %u of %u) is not usable!\n", device[k], k, n_gpus);
            exit(1);
        }
    }
    // joining a communicator is collective: one host thread per rank.  A join that does not come back cannot be
    // interrupted from outside; after RALA_JOIN_TIMEOUT seconds (default 300) the run ends with an error instead of
    // waiting for ever.
    double limit = 300.0;
    if (const char* e = getenv("RALA_JOIN_TIMEOUT")) limit = std::max(1.0, atof(e));
    struct Joined {
        std::mutex m;
        std::condition_variable cv;
        uint32_t done = 0;
        std::vector<int> rc;
    };
    auto joined = std::make_shared<Joined>();            // (outlives this function if a thread is stuck)
    joined->rc.assign(n_gpus, 0);
    const void* token = local ? local_group_ : (const void*)id;
    unsigned char* id_copy = nullptr;
    if (!local) {                                        // (the id, too)
        id_copy = new unsigned char[128];
        memcpy(id_copy, id, 128);
        token = id_copy;
    }
    for (uint32_t k = 0; k < n_gpus; ++k) {
        rala_hip_mg* mg = ranks_[k];
        std::thread([joined, mg, k, local, token]() {
            const int r = rala_hip_mg_join(mg, local ? RALA_HIP_COMM_LOCAL : RALA_HIP_COMM_RCCL, token);
            std::lock_guard<std::mutex> hold(joined->m);
            joined->rc[k] = r;
            ++joined->done;
            joined->cv.notify_all();
        }).detach();
    }
    {
        std::unique_lock<std::mutex> hold(joined->m);
        const bool all = joined->cv.wait_for(hold, std::chrono::duration<double>(limit), [&] { return joined->done == n_gpus; });
        if (!all) {
            fprintf(stderr, "[rala::Graph::Graph] error: %u of %u ranks did not join the %s group within %.0f s!\n",
                    n_gpus - joined->done, n_gpus, local ? "in-process" : "RCCL", limit);
            fflush(stderr);
            _exit(1);                                    // (threads are stuck inside the library: no orderly exit)
        }
        rc = joined->rc;
    }
    delete[] id_copy;
    for (uint32_t k = 0; k < n_gpus; ++k) {
        if (rc[k] != RALA_HIP_OK) {
            fprintf(stderr, "[rala::Graph::Graph] error: rank %u of %u could not join the group (%s)!\n", k, n_gpus,
                    rala_hip_mg_last_error(ranks_[k]));
            exit(1);
        }
    }
    ctx_ = rala_hip_mg_context(ranks_[0]);
}

Graph::~Graph() {
    piles_.clear();
    if (!ranks_.empty()) {
        for (rala_hip_mg* r : ranks_) rala_hip_mg_destroy(r);      // (they own ctx_)
        if (local_group_) rala_hip_mg_local_group_destroy(local_group_);
    } else if (ctx_) {
        rala_hip_destroy(ctx_);
    }
}

namespace {

bool read_sequences(const std::string& path, const io::SequenceSink& sink) {
    if (io::has_suffix(path, ".fastq") || io::has_suffix(path, ".fq") || io::has_suffix(path, ".fastq.gz") ||
        io::has_suffix(path, ".fq.gz")) {
        return io::read_fastq(path, sink);
    }
    return io::read_fasta(path, sink);
}

// one pass over an overlap file into binary columns (the reference parses twice, graph.cpp:328,443);
// uncompressed PAF goes through the multi-threaded reader.  check_lengths: the primary overlaps go
// through Overlap::transmute, which compares both lengths with the sequence file
// (overlap.cpp:54-59,73-78); the sensitive ones through transmute_, which checks nothing (:84-114)
void read_overlaps(const std::string& path, const std::unordered_map<std::string, uint64_t>& name_to_id,
    const io::NameTable& name_table, const std::vector<uint32_t>& read_len, bool check_lengths,
    uint32_t num_threads, io::OverlapColumns& c) {
    auto push = [&](uint32_t a, uint32_t b, uint32_t ab, uint32_t ae, uint32_t bb, uint32_t be, uint32_t len,
                    uint32_t strand) {
        c.a_id.push_back(a); c.b_id.push_back(b);
        c.a_begin.push_back(ab); c.a_end.push_back(ae);
        c.b_begin.push_back(bb); c.b_end.push_back(be);
        c.length.push_back(len); c.strand.push_back((uint8_t)strand);
    };
    auto length_error = [](uint64_t id) {
        fprintf(stderr, "[rala::Overlap::transmute] error: "
            "unequal lengths in sequence and overlap file for sequence with id %lu!\n", id);
        exit(1);
    };
    bool ok;
    int64_t bad = -1;
    if (io::has_suffix(path, ".paf")) {
        ok = io::read_paf_parallel(path, name_table, read_len, check_lengths, num_threads, c, &bad);
    } else if (io::has_suffix(path, ".paf.gz")) {
        // one thread inflates, the others parse (io.cpp)
        ok = io::read_overlaps_streamed(path, false, name_table, read_len, check_lengths, std::max(2u, num_threads), c, &bad);
    } else if (io::has_suffix(path, ".mhap") || io::has_suffix(path, ".mhap.gz")) {
        ok = io::read_overlaps_streamed(path, true, name_table, read_len, check_lengths, std::max(2u, num_threads), c, &bad);
    } else {
        ok = io::read_paf(path, [&](const io::PafRecord& r) {
            auto a = name_to_id.find(r.q_name), b = name_to_id.find(r.t_name);
            const uint32_t ia = a == name_to_id.end() ? RALA_HIP_NO_READ : (uint32_t)a->second;
            const uint32_t ib = b == name_to_id.end() ? RALA_HIP_NO_READ : (uint32_t)b->second;
            // Overlap::transmute checks a first and stops at the first unknown name (overlap.cpp:43-78)
            if (check_lengths && ia != RALA_HIP_NO_READ && r.q_length != read_len[ia]) length_error(ia);
            if (check_lengths && ia != RALA_HIP_NO_READ && ib != RALA_HIP_NO_READ && r.t_length != read_len[ib]) {
                length_error(ib);
            }
            push(ia, ib, r.q_begin, r.q_end, r.t_begin, r.t_end, r.overlap_length, r.orientation == '+' ? 0 : 1);
        });
    }
    if (ok && bad >= 0) length_error((uint64_t)bad);
    if (!ok) {
        fprintf(stderr, "[rala::Graph::initialize] error: unable to open file %s!\n", path.c_str());
        exit(1);
    }
}

}  // namespace

// reference src/graph.cpp:244-425
void Graph::initialize() {
    StageTimer timer;
    if (!read_sequences(sequences_path_, [&](const std::string& name, const std::string& data) {
            name_to_id_[name] = names_.size();
            names_.push_back(name);
            read_len_.push_back((uint32_t)data.size());
        })) {
        fprintf(stderr, "[rala::Graph::initialize] error: unable to open file %s!\n", sequences_path_.c_str());
        exit(1);
    }
    timer("[rala::Graph::initialize] loaded sequences");
    timer();
    name_table_.build(names_);
    // An uncompressed PAF file: the text goes to the device and is tokenised there (rala_hip_set_overlaps_from_paf;
    // RALA_DEVICE_INGEST=0 keeps the host reader).  A file that tokeniser calls irregular - or cannot take: no regular file,
    // too large for its 32-bit counts, no room for its text in device memory - falls through to the host reader, which knows
    // what to do with such files (a FIFO named *.paf used to end with "unable to open file").
    // (round 6: an uncompressed MHAP file too - twelve numeric columns, no names to look up; one GPU - several take the host reader)
    const bool mhap_text = io::has_suffix(overlaps_path_, ".mhap");
    const bool device_ingest = (io::has_suffix(overlaps_path_, ".paf") || (mhap_text && ranks_.empty())) &&
                               !(getenv("RALA_DEVICE_INGEST") && atoi(getenv("RALA_DEVICE_INGEST")) == 0);
    auto falls_back = [](int rc) { return rc == RALA_HIP_ENOTAFILE || rc == RALA_HIP_ETOOLARGE || rc == RALA_HIP_ENOMEM; };
    auto length_error = [](int64_t bad) {
        fprintf(stderr, "[rala::Overlap::transmute] error: "
            "unequal lengths in sequence and overlap file for sequence with id %lu!\n", (uint64_t)bad);
        exit(1);
    };
    if (ranks_.empty() && device_ingest) {
        check(ctx_, rala_hip_set_reads(ctx_, read_len_.data(), read_len_.size()), "initialize");
        check(ctx_, rala_hip_set_name_table(ctx_, name_table_.buckets(), name_table_.n_buckets(), name_table_.arena().data(),
                                            name_table_.arena().size()), "initialize");
        int64_t bad = -1;
        int irregular = 0;
        const int rc = mhap_text ? rala_hip_set_overlaps_from_mhap(ctx_, overlaps_path_.c_str(), 1, std::max(1u, num_threads_), &bad, &irregular)
                                 : rala_hip_set_overlaps_from_paf(ctx_, overlaps_path_.c_str(), 1, std::max(1u, num_threads_), &bad, &irregular);
        if (rc == RALA_HIP_EINVAL && !irregular) {
            // (cannot open: the reference's message)
            fprintf(stderr, "[rala::Graph::initialize] error: unable to open file %s!\n", overlaps_path_.c_str());
            exit(1);
        }
        if (!falls_back(rc)) {
            check(ctx_, rc, "initialize");
            if (bad >= 0) length_error(bad);
            if (!irregular) {
                timer("[rala::Graph::initialize] loaded overlaps");
                timer();
                initialize_piles();
                return;
            }
        }
    }
    if (!ranks_.empty() && device_ingest) {
        // several GPUs: every rank ships and tokenises its own byte range of the file on its own GPU, the ranks settle the
        // cuts between runs of equal queries among themselves (rala_hip_mg_set_overlaps_from_paf; round 5 - before, the host
        // parsed the whole file and handed every rank a slice from host memory)
        const uint32_t P = (uint32_t)ranks_.size();
        std::vector<int> rc(P, RALA_HIP_OK), irregular(P, 0);
        std::vector<int64_t> bad(P, -1);
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < P; ++k) {
            th.emplace_back([&, k]() {
                rala_hip_mg* mg = ranks_[k];
                int r = rala_hip_mg_set_reads(mg, read_len_.data(), read_len_.size());
                if (r == RALA_HIP_OK) {
                    r = rala_hip_set_name_table(rala_hip_mg_context(mg), name_table_.buckets(), name_table_.n_buckets(),
                                                name_table_.arena().data(), name_table_.arena().size());
                }
                // (collective: a rank that failed above still calls it, with nothing to read, so that nobody waits for it)
                const int r2 = rala_hip_mg_set_overlaps_from_paf(mg, r == RALA_HIP_OK ? overlaps_path_.c_str() : "", 1,
                                                                 std::max(1u, num_threads_ / P), &bad[k], &irregular[k]);
                rc[k] = r != RALA_HIP_OK ? r : r2;
            });
        }
        for (auto& t : th) t.join();
        bool every_rank_ok = true, fall = false;
        for (uint32_t k = 0; k < P; ++k) {
            if (rc[k] == RALA_HIP_EINVAL && access(overlaps_path_.c_str(), R_OK) != 0) {
                fprintf(stderr, "[rala::Graph::initialize] error: unable to open file %s!\n", overlaps_path_.c_str());
                exit(1);
            }
            every_rank_ok = every_rank_ok && rc[k] == RALA_HIP_OK;
            fall = fall || falls_back(rc[k]) || irregular[k];
        }
        if (every_rank_ok && !fall) {
            for (uint32_t k = 0; k < P; ++k) if (bad[k] >= 0) length_error(bad[k]);
            timer("[rala::Graph::initialize] loaded overlaps");
            return;
        }
        if (!fall) {
            for (uint32_t k = 0; k < P; ++k) {
                if (rc[k] != RALA_HIP_OK) fprintf(stderr, "[rala::Graph::initialize] error: %s!\n", rala_hip_mg_last_error(ranks_[k]));
            }
            exit(1);
        }
        // (irregular, or beyond the tokeniser's limits on some rank: the host reader below - the group is intact unless a
        // rank failed alone, which ends the run at the first collective)
    }
    read_overlaps(overlaps_path_, name_to_id_, name_table_, read_len_, true, num_threads_, overlaps_);
    if (!ranks_.empty()) {
        // every rank gets all read lengths and its slice of the overlaps (cut between runs of
        // equal queries); the stages themselves run in construct(), all ranks together
        const uint32_t P = (uint32_t)ranks_.size();
        std::vector<uint64_t> cuts(P + 1);
        rala_hip_mg_slice_cuts(overlaps_.a_id.data(), overlaps_.b_id.data(), overlaps_.size(), P, cuts.data());
        for (uint32_t k = 0; k < P; ++k) {
            const uint64_t lo = cuts[k];
            rala_hip_overlaps sl = {overlaps_.a_id.data() + lo, overlaps_.b_id.data() + lo, overlaps_.a_begin.data() + lo,
                                    overlaps_.a_end.data() + lo, overlaps_.b_begin.data() + lo, overlaps_.b_end.data() + lo,
                                    overlaps_.length.data() + lo, overlaps_.strand.data() + lo};
            if (rala_hip_mg_set_reads(ranks_[k], read_len_.data(), read_len_.size()) != RALA_HIP_OK ||
                rala_hip_mg_set_overlaps(ranks_[k], &sl, cuts[k + 1] - lo, lo, RALA_HIP_MEM_HOST) != RALA_HIP_OK) {
                fprintf(stderr, "[rala::Graph::initialize] error: %s!\n", rala_hip_mg_last_error(ranks_[k]));
                exit(1);
            }
        }
        timer("[rala::Graph::initialize] loaded overlaps");
        return;
    }
    check(ctx_, rala_hip_set_reads(ctx_, read_len_.data(), read_len_.size()), "initialize");
    rala_hip_overlaps soa = {overlaps_.a_id.data(), overlaps_.b_id.data(), overlaps_.a_begin.data(),
                             overlaps_.a_end.data(), overlaps_.b_begin.data(), overlaps_.b_end.data(),
                             overlaps_.length.data(), overlaps_.strand.data()};
    check(ctx_, rala_hip_set_overlaps(ctx_, &soa, overlaps_.size(), RALA_HIP_MEM_HOST), "initialize");
    timer("[rala::Graph::initialize] loaded overlaps");
    initialize_piles();
}

// the second half of Graph::initialize (graph.cpp:384-425) once the overlaps are on the device
void Graph::initialize_piles() {
    StageTimer timer;
    const int rc = rala_hip_initialize(ctx_);
    timer("[rala::Graph::initialize] prefiltered sequences");
    if (rc == RALA_HIP_EFILTERED) {
        fprintf(stderr, "[rala::Graph::initialize] error: filtered all sequences!\n");
        exit(1);
    }
    check(ctx_, rc, "initialize");
    uint64_t num_prefiltered_sequences = 0;
    rala_hip_get_num_prefiltered(ctx_, &num_prefiltered_sequences);
    fprintf(stderr, "[rala::Graph::initialize] number of prefiltered sequences = %lu\n", num_prefiltered_sequences);
}

// reference src/graph.cpp:427-640
void Graph::construct(const std::string& sensitive_overlaps_path) {
    if (!piles_.empty()) {
        fprintf(stderr, "[rala::Graph::construct] warning: object already constructed!\n");
        return;
    }
    initialize();

    StageTimer timer;
    io::OverlapColumns s_cols;
    rala_hip_overlaps sens = {};
    uint64_t n_sens = 0;
    // the sensitive overlaps tokenised on the device(s): every context's share as device pointers
    bool sens_on_device_ = false;
    std::vector<rala_hip_overlaps> device_shares_;
    std::vector<uint64_t> device_share_n_;
    if (!sensitive_overlaps_path.empty()) {
        if (!(io::has_suffix(sensitive_overlaps_path, ".mhap") || io::has_suffix(sensitive_overlaps_path, ".mhap.gz") ||
              io::has_suffix(sensitive_overlaps_path, ".paf") || io::has_suffix(sensitive_overlaps_path, ".paf.gz"))) {
            fprintf(stderr, "[rala::preprocess] error: "
                "file %s has unsupported format extension (valid extensions: "
                ".mhap, .mhap.gz, .paf, .paf.gz)!\n", sensitive_overlaps_path.c_str());
            exit(1);
        }
        // an uncompressed PAF file is tokenised on the device(s) - no length check, Overlap::transmute_ has none (round 5;
        // RALA_DEVICE_INGEST=0, or a file the tokeniser cannot take: the host reader)
        const bool device_ingest = io::has_suffix(sensitive_overlaps_path, ".paf") && name_table_.n_buckets() != 0 &&
                                   !(getenv("RALA_DEVICE_INGEST") && atoi(getenv("RALA_DEVICE_INGEST")) == 0);
        if (device_ingest) {
            const uint32_t P = ranks_.empty() ? 1u : (uint32_t)ranks_.size();
            uint64_t file_bytes = 0;
            {
                std::ifstream f(sensitive_overlaps_path, std::ios::binary | std::ios::ate);
                if (f) file_bytes = (uint64_t)f.tellg();
            }
            device_shares_.assign(P, rala_hip_overlaps());
            device_share_n_.assign(P, 0);
            std::vector<int> rc(P, RALA_HIP_OK), irregular(P, 0);
            std::vector<std::thread> th;
            for (uint32_t k = 0; k < P; ++k) {
                th.emplace_back([&, k]() {
                    rala_hip_ctx* c = ranks_.empty() ? ctx_ : rala_hip_mg_context(ranks_[k]);
                    // (a run that took the host reader for the primary overlaps has set no name table yet)
                    rc[k] = rala_hip_set_name_table(c, name_table_.buckets(), name_table_.n_buckets(), name_table_.arena().data(),
                                                    name_table_.arena().size());
                    if (rc[k] == RALA_HIP_OK) {
                        rc[k] = rala_hip_tokenise_sensitive_paf(c, sensitive_overlaps_path.c_str(), file_bytes / P * k + std::min<uint64_t>(k, file_bytes % P),
                                                                file_bytes / P * (k + 1) + std::min<uint64_t>(k + 1, file_bytes % P),
                                                                std::max(1u, num_threads_ / P), &device_shares_[k], &device_share_n_[k], &irregular[k]);
                    }
                });
            }
            for (auto& t : th) t.join();
            sens_on_device_ = true;
            for (uint32_t k = 0; k < P; ++k) sens_on_device_ = sens_on_device_ && rc[k] == RALA_HIP_OK && !irregular[k];
            if (sens_on_device_) {
                for (uint32_t k = 0; k < P; ++k) {
                    rala_hip_ctx* c = ranks_.empty() ? ctx_ : rala_hip_mg_context(ranks_[k]);
                    check(c, rala_hip_set_option(c, "sensitive_in_device_memory", 1), "construct");
                    n_sens += device_share_n_[k];
                }
                if (ranks_.empty()) sens = device_shares_[0];
            }
        }
        if (!sens_on_device_) {
            read_overlaps(sensitive_overlaps_path, name_to_id_, name_table_, read_len_, false, num_threads_, s_cols);
            sens.a_id = s_cols.a_id.data(); sens.b_id = s_cols.b_id.data(); sens.a_begin = s_cols.a_begin.data();
            sens.a_end = s_cols.a_end.data(); sens.b_begin = s_cols.b_begin.data(); sens.b_end = s_cols.b_end.data();
            sens.length = s_cols.length.data(); sens.strand = s_cols.strand.data();
            n_sens = s_cols.size();
        }
    }
    if (!ranks_.empty()) {
        // all ranks together: piles on the owners, filtering per slice, tail + graph replicated
        const uint32_t P = (uint32_t)ranks_.size();
        std::vector<rala_hip_overlaps> shares(P);
        std::vector<uint64_t> n_share(P, 0);
        for (uint32_t k = 0; k < P; ++k) {
            if (sens_on_device_) {
                shares[k] = device_shares_[k];
                n_share[k] = device_share_n_[k];
                continue;
            }
            const uint64_t lo = n_sens * k / P, hi = n_sens * (k + 1) / P;
            shares[k] = {sens.a_id + lo, sens.b_id + lo, sens.a_begin + lo, sens.a_end + lo, sens.b_begin + lo,
                         sens.b_end + lo, sens.length + lo, sens.strand + lo};
            n_share[k] = hi - lo;
        }
        uint32_t pairs = 0;
        const bool with_sens = !sensitive_overlaps_path.empty() && (n_sens != 0 || sens_on_device_);
        const int rc = rala_hip_mg_run_threads(ranks_.data(), P, with_sens ? shares.data() : nullptr, with_sens ? n_share.data() : nullptr,
                                               &pairs);
        if (rc == RALA_HIP_EFILTERED) {
            fprintf(stderr, "[rala::Graph::initialize] error: filtered all sequences!\n");
            exit(1);
        }
        if (rc != RALA_HIP_OK) {
            for (rala_hip_mg* r : ranks_) {
                if (rala_hip_mg_last_error(r)[0]) fprintf(stderr, "[rala::Graph::construct] error: %s!\n", rala_hip_mg_last_error(r));
            }
            exit(1);
        }
        uint64_t num_prefiltered_sequences = 0;
        rala_hip_get_num_prefiltered(ctx_, &num_prefiltered_sequences);
        fprintf(stderr, "[rala::Graph::initialize] number of prefiltered sequences = %lu\n", num_prefiltered_sequences);
    } else {
        check(ctx_, rala_hip_construct(ctx_, n_sens ? &sens : nullptr, n_sens), "construct");
    }
    timer("[rala::Graph::construct] loaded overlaps + [rala::Graph::preprocess]");
    timer();

    // piles_: views of the surviving reads
    const uint64_t n = read_len_.size();
    std::vector<uint32_t> begin(n), end(n);
    std::vector<uint16_t> median(n), p10(n);
    std::vector<uint8_t> alive(n);
    rala_hip_get_piles(ctx_, begin.data(), end.data(), median.data(), p10.data(), alive.data());
    std::vector<uint64_t> off[3];
    std::vector<uint32_t> pairs[3], aux[3];
    for (int kind = 0; kind < 3; ++kind) {
        off[kind].resize(n + 1);
        rala_hip_get_intervals(ctx_, kind, off[kind].data(), nullptr, nullptr);
        pairs[kind].resize(2 * off[kind][n] + 2);
        aux[kind].resize(off[kind][n] + 1);
        rala_hip_get_intervals(ctx_, kind, off[kind].data(), pairs[kind].data(), aux[kind].data());
    }
    piles_.resize(n);
    for (uint64_t r = 0; r < n; ++r) {
        if (!alive[r]) continue;
        piles_[r] = createPile(r, read_len_[r]);
        Pile& p = *piles_[r];
        p.ctx_ = ctx_; p.owns_ctx_ = false; p.ctx_read_ = r; p.computed_ = true;
        if (!ranks_.empty()) p.mg_ = ranks_[r % ranks_.size()];         // the coverage lives on the owner
        p.begin_ = begin[r]; p.end_ = end[r]; p.median_ = median[r]; p.p10_ = p10[r];
        for (uint64_t k = off[0][r]; k < off[0][r + 1]; ++k) {
            p.chimeric_pits_.emplace_back(pairs[0][2 * k], pairs[0][2 * k + 1]);
            p.chimeric_pit_min_.push_back((uint16_t)aux[0][k]);
        }
        for (uint64_t k = off[2][r]; k < off[2][r + 1]; ++k) {
            p.repeat_hills_.emplace_back(pairs[2][2 * k], pairs[2][2 * k + 1]);
            p.repeat_hill_coverage_.push_back(aux[2][k] != 0);
        }
    }

    // nodes (second pass over the sequences, trimmed) and edges (graph.cpp:527-632)
    uint64_t n_nodes = 0, n_edges = 0;
    rala_hip_get_graph_size(ctx_, &n_nodes, &n_edges);
    std::vector<uint32_t> node_read(n_nodes), src(n_edges), dst(n_edges), len(n_edges);
    rala_hip_get_graph(ctx_, node_read.data(), src.data(), dst.data(), len.data(), nullptr);
    std::vector<int64_t> read_to_node(n, -1);
    for (uint64_t k = 0; k < n_nodes; k += 2) read_to_node[node_read[k]] = (int64_t)k;
    // nodes in read order (graph.cpp:553-574): collect the trimmed sequences first
    std::vector<std::string> node_name(n_nodes / 2), node_data(n_nodes / 2), node_rc(n_nodes / 2);
    uint64_t seq_id = 0;
    read_sequences(sequences_path_, [&](const std::string& name, const std::string& data) {
        const uint64_t i = seq_id++;
        if (i >= n || read_to_node[i] < 0) return;
        auto seq = createSequence(name, data);
        seq->trim(begin[i], end[i]);
        const uint64_t k = (uint64_t)read_to_node[i] / 2;
        node_name[k] = name;
        node_data[k] = seq->data();
        node_rc[k] = seq->reverse_complement();
    });
    for (uint64_t k = 0; k < n_nodes / 2; ++k) {
        graph_.add_sequence_nodes(node_read[2 * k], node_name[k], node_data[k], node_rc[k]);
    }
    timer("[rala::Graph::construct] loaded sequences");
    timer();
    for (uint64_t e = 0; e < n_edges; ++e) graph_.add_edge(src[e], dst[e], len[e]);
    timer("[rala::Graph::construct] created assembly graph");
    fprintf(stderr, "[rala::Graph::construct] number of nodes = %zu\n", graph_.nodes().size());
    fprintf(stderr, "[rala::Graph::construct] number of edges = %zu\n", graph_.edges().size());
}

// reference src/graph.cpp:642-697; the layout of postprocess() runs on the GPU with a fixed seed
// (the reference seeds it from std::random_device, so its long-edge decisions vary run to run)
void Graph::simplify() {
    StageTimer timer;
    const uint32_t num_transitive_edges = remove_transitive_edges();
    uint32_t num_tips = 0, num_bubbles = 0, num_long_edges = 0;
    auto tips_and_bubbles = [&]() {
        while (true) {
            uint32_t num_changes = remove_tips();
            num_tips += num_changes;
            const uint32_t num_changes_part = remove_bubbles();
            num_bubbles += num_changes_part;
            num_changes += num_changes_part;
            if (num_changes == 0) break;
        }
    };
    tips_and_bubbles();
    shrink(42);
    for (uint32_t i = 0; i < 5; ++i) {
        postprocess();
        num_long_edges += remove_long_edges();
        num_tips += remove_tips();
    }
    tips_and_bubbles();
    timer("[rala::Graph::simplify]");
    fprintf(stderr, "[rala::Graph::simplify] number of transitive edges = %u\n", num_transitive_edges);
    fprintf(stderr, "[rala::Graph::simplify] number of tips = %u\n", num_tips);
    fprintf(stderr, "[rala::Graph::simplify] number of bubbles = %u\n", num_bubbles);
    fprintf(stderr, "[rala::Graph::simplify] number of long edges = %u\n", num_long_edges);
}

// reference src/graph.cpp:1281-1335 through rala_hip_tr_mark on the current live edges
uint32_t Graph::remove_transitive_edges() {
    const auto& edges = graph_.edges();
    std::vector<uint32_t> ids, src, dst, len;
    for (uint64_t e = 0; e < edges.size(); ++e) {
        if (!edges[e].alive) continue;
        ids.push_back((uint32_t)e);
        src.push_back(edges[e].begin_node); dst.push_back(edges[e].end_node); len.push_back(edges[e].length);
    }
    std::vector<uint8_t> marks(ids.size());
    uint32_t num_transitive_edges = 0;
    check(ctx_, rala_hip_tr_mark(ctx_, (uint32_t)graph_.nodes().size(), (uint32_t)ids.size(), src.data(), dst.data(),
                                 len.data(), marks.data(), &num_transitive_edges), "remove_transitive_edges");
    // marks come in twin pairs (graph.cpp:1306-1309)
    for (size_t k = 0; k < ids.size(); ++k) {
        if (marks[k] && !graph_.edges()[ids[k]].is_marked) graph_.mark_edge(ids[k]);
    }
    graph_.note_transitive_edges();
    graph_.remove_marked_objects();
    return num_transitive_edges;
}

// reference src/graph.cpp:1056-1279
void Graph::postprocess() {
    StageTimer timer;
    const int rc = graph_.postprocess([&](uint32_t n, double* x, double* y, const uint32_t* adj_off, const uint32_t* adj,
                                          uint32_t iterations, double k, double t, double dt) {
        return rala_hip_layout(ctx_, n, x, y, adj_off, adj, iterations, k, t, dt);
    }, layout_seed_++);
    check(ctx_, rc, "postprocess");
    timer("[rala::Graph::postprocess]");
}

uint32_t Graph::remove_long_edges() { return graph_.remove_long_edges(); }
uint32_t Graph::remove_tips() { return graph_.remove_tips(); }
uint32_t Graph::remove_bubbles() { return graph_.remove_bubbles(); }
uint32_t Graph::create_unitigs() { return graph_.create_unitigs(); }
uint32_t Graph::shrink(uint32_t epsilon) { return graph_.shrink(epsilon); }

// reference src/graph.cpp:2042-2082
void Graph::extract_contigs(std::vector<std::unique_ptr<Sequence>>& dst, bool drop_unassembled_sequences) {
    create_unitigs();
    uint32_t contig_id = 0;
    std::vector<uint32_t> contig_length;
    for (const auto& node : graph_.nodes()) {
        if (!node.alive || node.is_rc()) continue;
        if (drop_unassembled_sequences && (node.sequence_ids.size() < 6 || node.length() < 10000)) continue;
        contig_length.push_back(node.length());
        std::string name = "Ctg" + std::to_string(contig_id);
        name += " RC:i:" + std::to_string(node.sequence_ids.size());
        name += " LN:i:" + std::to_string(node.data.size());
        dst.emplace_back(createSequence(name, node.data));
        ++contig_id;
    }
    fprintf(stderr, "[rala::Graph::extract_contigs] number of contigs = %zu\n", contig_length.size());
    if (contig_length.empty()) return;
    std::sort(contig_length.begin(), contig_length.end());
    fprintf(stderr, "[rala::Graph::extract_contigs] shortest contig length = %u\n", contig_length.front());
    fprintf(stderr, "[rala::Graph::extract_contigs] median contig length = %u\n",
        contig_length[contig_length.size() / 2]);
    fprintf(stderr, "[rala::Graph::extract_contigs] longest contig length = %u\n", contig_length.back());
}

// reference src/graph.cpp:2084-2116 (the reference iterates an unordered_set; here ascending ids)
void Graph::extract_nodes(std::vector<std::unique_ptr<Sequence>>& dst) {
    const auto& nodes = graph_.nodes();
    const auto& edges = graph_.edges();
    std::set<uint64_t> node_ids;
    for (const auto& it : nodes) {
        if (!it.alive || it.is_rc() || (it.outdegree() == 0 && it.indegree() == 0)) continue;
        node_ids.insert(it.id);
        for (uint32_t e : it.prefix_edges) node_ids.insert(edges[e].begin_node & ~1u);
        for (uint32_t e : it.suffix_edges) node_ids.insert(edges[e].end_node & ~1u);
    }
    for (uint64_t id : node_ids) dst.emplace_back(createSequence(nodes[id].name, nodes[id].data));
    fprintf(stderr, "[rala::Graph::extract_nodes] number of nodes = %zu\n", dst.size());
}

// reference src/graph.cpp:2153-2297: the writers live in AssemblyGraph (assembly_graph.cpp)
void Graph::print_csv(const std::string& path) const {
    auto graph_file = fopen(path.c_str(), "w");
    if (!graph_file) return;
    graph_.write_csv(graph_file);
    fclose(graph_file);
}

void Graph::print_gfa(const std::string& path) const {
    auto graph_file = fopen(path.c_str(), "w");
    if (!graph_file) return;
    graph_.write_gfa(graph_file);
    fclose(graph_file);
}

void Graph::print_json(const std::string& path) const {
    std::ofstream os(path);
    graph_.write_json(os, [&](uint64_t id) { return piles_[id] == nullptr ? std::string() : piles_[id]->to_json(); });
}

void Graph::print_debug(const std::string& prefix) const {
    if (!prefix.empty()) {
        print_csv(prefix + ".csv");
        print_json(prefix + ".json");
    }
}

}  // namespace rala

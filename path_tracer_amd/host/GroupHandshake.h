// GroupHandshake.h — how the ranks of `pt_main --group RANK WORLD IDFILE` pass the RCCL id around (no reference
// counterpart: the reference is one process on one GPU, src/Main.cpp:9).  Files only, so it needs nothing but a
// directory all ranks can reach:
//   rank r > 0  writes  IDFILE.req<r>  = a nonce of its own (pid, clock, address entropy), then polls IDFILE;
//   rank 0      removes any IDFILE left by an earlier run, waits for all WORLD-1 request files, then publishes
//               IDFILE = [id bytes][nonce of rank 1] ... [nonce of rank WORLD-1]  (written aside, renamed: never half-seen),
//               again whenever a request file changes, until every request file has been taken away;
//   rank r > 0  accepts IDFILE only if slot r holds ITS nonce -- a stale file from a run that died cannot match --
//               and then removes its request file; rank 0 removes IDFILE once the group exists (handshake_cleanup).
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

namespace ptm {

inline uint64_t handshake_nonce()
{
    uint64_t x = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    x ^= (uint64_t)getpid() * 0x9e3779b97f4a7c15ull;
    x ^= (uint64_t)(uintptr_t)&x << 17;
    x ^= (uint64_t)std::chrono::system_clock::now().time_since_epoch().count() * 0xbf58476d1ce4e5b9ull;
    return x ? x : 1;
}

inline bool handshake_read_all(const std::string &path, void *buf, size_t n)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool ok = std::fread(buf, 1, n, f) == n;
    std::fclose(f);
    return ok;
}

inline bool handshake_write_atomically(const std::string &path, const void *buf, size_t n)
{
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) { std::perror(tmp.c_str()); return false; }
    const bool ok = std::fwrite(buf, 1, n, f) == n;
    if (std::fclose(f) != 0 || !ok) { std::perror(tmp.c_str()); std::remove(tmp.c_str()); return false; }
    if (std::rename(tmp.c_str(), path.c_str())) { std::perror(path.c_str()); std::remove(tmp.c_str()); return false; }
    return true;
}

// Rank 0 passes the id it made in `id`; the other ranks receive it there.  `make_id` is only called on rank 0, after the
// stale file is gone.  Returns false (with a message on stderr) after `timeout_s` seconds without the peers.
template <class MakeId>
inline bool handshake_exchange(int rank, int world, const std::string &idfile, char *id, size_t id_bytes, MakeId make_id, double timeout_s = 60.0)
{
    const auto t0 = std::chrono::steady_clock::now();
    auto expired = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
    auto nap = []() { std::this_thread::sleep_for(std::chrono::milliseconds(20)); };
    const size_t total = id_bytes + sizeof(uint64_t) * (size_t)(world > 1 ? world - 1 : 0);
    if (rank == 0) {
        std::remove(idfile.c_str());
        if (!make_id(id)) return false;
        std::vector<char> blob(total);
        std::memcpy(blob.data(), id, id_bytes);
        // Publish as soon as every rank has asked; publish again whenever a request changes (a request file left by a run
        // that died is overwritten by the live rank a moment later); done when every rank has taken its request file away,
        // which it does only after reading ITS nonce back from IDFILE.
        std::vector<uint64_t> asked((size_t)world, 0);
        std::vector<char> acked((size_t)world, 0);
        for (;;) {
            bool changed = false, all_asked = true, all_acked = true;
            for (int r = 1; r < world; r++) {
                if (acked[r]) continue;
                uint64_t nonce = 0;
                const std::string req = idfile + ".req" + std::to_string(r);
                if (handshake_read_all(req, &nonce, sizeof nonce) && nonce != 0) {
                    if (nonce != asked[r]) { asked[r] = nonce; changed = true; }
                } else if (asked[r] != 0 && access(req.c_str(), F_OK) != 0) {
                    acked[r] = 1;                       // published for it before, and the file is gone: rank r has its id
                }
                if (!asked[r]) all_asked = false;
                if (!acked[r]) all_acked = false;
            }
            if (all_asked && changed) {
                for (int r = 1; r < world; r++) std::memcpy(blob.data() + id_bytes + sizeof(uint64_t) * (size_t)(r - 1), &asked[r], sizeof(uint64_t));
                if (!handshake_write_atomically(idfile, blob.data(), total)) return false;
            }
            if (all_acked) return true;
            if (expired()) { std::fprintf(stderr, "group handshake: not every rank took the id within %.0f s\n", timeout_s); return false; }
            nap();
        }
    }
    const uint64_t mine = handshake_nonce();
    const std::string req = idfile + ".req" + std::to_string(rank);
    if (!handshake_write_atomically(req, &mine, sizeof mine)) return false;
    std::vector<char> blob(total);
    for (;;) {
        uint64_t got = 0;
        if (handshake_read_all(idfile, blob.data(), total)) std::memcpy(&got, blob.data() + id_bytes + sizeof(uint64_t) * (size_t)(rank - 1), sizeof got);
        if (got == mine) break;
        if (expired()) { std::fprintf(stderr, "group handshake: no id for rank %d in %s after %.0f s\n", rank, idfile.c_str(), timeout_s); std::remove(req.c_str()); return false; }
        nap();
    }
    std::memcpy(id, blob.data(), id_bytes);
    std::remove(req.c_str());
    return true;
}

// after the group exists (pt_group_init is collective, so every rank has read the file by then)
inline void handshake_cleanup(int rank, const std::string &idfile) { if (rank == 0) std::remove(idfile.c_str()); }

}  // namespace ptm

// TEST INFRASTRUCTURE.  The work queue of the persistent kernels (csrc/vxrt_device.hpp: queue_ticket, queue_holds,
// queue_next_shard, and the control flow of queue_take) on the host: every ticket of a queue belongs to exactly one shard,
// and waves that take tickets in any interleaving hand out every ticket once and all leave.
// usage: queue_check            (exit code 0 = all good; prints one line per failure)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define VXRT_HOST_CHECK 1
#include "../../voxelengine_amd/csrc/vxrt_device.hpp"

using namespace vxrt;

static int failures = 0;

static void check_partition(uint32_t total, uint32_t granule)
{
    std::vector<uint8_t> seen(total, 0);
    uint64_t sum = 0;
    for (uint32_t q = 0; q < kQueueShards; ++q) {
        const uint32_t holds = queue_holds(q, total, granule);
        sum += holds;
        for (uint32_t l = 0; l < holds; ++l) {
            const uint32_t t = queue_ticket(q, l, granule);
            if (t >= total || seen[t]++) { printf("partition: total %u granule %u shard %u local %u -> ticket %u\n", total, granule, q, l, t); ++failures; return; }
        }
        // the first local number past the shard's end maps past the queue's end (what makes `local < holds` the validity test)
        if (queue_ticket(q, holds, granule) < total && queue_holds(q, total, granule) == holds) {
            // (a ticket below `total` there would belong to this shard too: then holds was too small)
            printf("partition: total %u granule %u shard %u: local %u is still a ticket of the queue\n", total, granule, q, holds); ++failures; return;
        }
    }
    if (sum != total) { printf("partition: total %u granule %u: shards hold %llu tickets\n", total, granule, (unsigned long long)sum); ++failures; }
}

// queue_take's control flow with the counters in a host array; `waves` waves step in a random interleaving, one memory
// operation (the atomic, or the load of all heads) per step, so a wave's view of the heads can be stale as on the device
static void check_waves(uint32_t total, uint32_t granule, uint32_t waves, unsigned seed)
{
    std::vector<uint32_t> heads(kQueueShards, 0u), got(total, 0u);
    struct Wave { uint32_t shard; int state; uint32_t open; bool done; uint32_t taken; };  // state 0: atomic next, 1: heads loaded
    std::vector<Wave> w(waves);
    for (uint32_t i = 0; i < waves; ++i) w[i] = {i % kQueueShards, 0, 0u, false, 0u};
    srand(seed);
    uint32_t live = waves;
    uint64_t steps = 0;
    while (live) {
        Wave& v = w[(uint32_t)rand() % waves];
        if (v.done) continue;
        if (++steps > 64ull * (total + 64ull * waves) + 1000000ull) { printf("waves: total %u granule %u waves %u: no end\n", total, granule, waves); ++failures; return; }
        if (v.state == 0) {
            const uint32_t local = heads[v.shard]++;
            if (local < queue_holds(v.shard, total, granule)) {
                const uint32_t t = queue_ticket(v.shard, local, granule);
                if (t >= total || got[t]++) { printf("waves: ticket %u handed out twice or out of range\n", t); ++failures; return; }
                ++v.taken;
            } else if (kQueueShards == 1u) {
                v.done = true; --live;
            } else {
                v.open = 0u;
                for (uint32_t k = 0; k < kQueueShards; ++k)
                    if (heads[k] < queue_holds(k, total, granule)) v.open |= 1u << k;
                v.state = 1;  // (the decision on this snapshot happens at the wave's next step: others move in between)
            }
        } else {
            if (v.open == 0u) { v.done = true; --live; }
            else { v.shard = queue_next_shard(v.open, v.shard); v.state = 0; }
        }
    }
    for (uint32_t t = 0; t < total; ++t)
        if (got[t] != 1u) { printf("waves: total %u granule %u waves %u: ticket %u handed out %u times\n", total, granule, waves, t, got[t]); ++failures; return; }
}

int main()
{
    const uint32_t totals[] = {0u, 1u, 2u, 7u, 8u, 9u, 31u, 32u, 33u, 63u, 64u, 65u, 255u, 1000u, 4097u, 32400u, 129600u};
    const uint32_t granules[] = {1u, 2u, 4u, 8u, 30u, 240u};
    for (uint32_t total : totals)
        for (uint32_t g : granules)
            check_partition(total, g);
    srand(7);
    for (int i = 0; i < 300; ++i)
        check_partition((uint32_t)rand() % 70000u, 1u + (uint32_t)rand() % 64u);
    for (uint32_t total : {0u, 1u, 5u, 64u, 1000u, 5000u})
        for (uint32_t g : {1u, 4u, 30u})
            for (uint32_t waves : {1u, 3u, 8u, 64u, 500u})
                check_waves(total, g, waves, total + 31u * g + waves);
    printf("queue_check: shards %u, %d failure(s)\n", kQueueShards, failures);
    return failures ? 1 : 0;
}

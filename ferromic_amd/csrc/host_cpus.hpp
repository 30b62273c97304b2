// Host CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (cgroup v2 cpu.max, v1
// cpu.cfs_quota_us / cpu.cfs_period_us).  std::thread::hardware_concurrency() is the machine: in a container with a 16-CPU quota on a
// 256-thread host, 64 worker threads burn the quota in 5-ms slices per runqueue and the whole process is throttled for the rest of every
// 100-ms period (measured: run_vcf's text ingest 1.66 s with 64 threads, blocks alternating between 10 and 80 ms).
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <thread>

namespace fmh_host {

inline unsigned usable_cpus() {
  static const unsigned n = [] {
    unsigned cpus = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) cpus = (unsigned)CPU_COUNT(&set);
    if (cpus == 0) cpus = 1;
    double quota = 0.0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      double period = 0.0;
      if (fscanf(f, "%31s %lf", q, &period) == 2 && q[0] != 'm' && period > 0.0) quota = atof(q) / period;
      fclose(f);
    } else {
      long long q = -1, period = 0;
      if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &q) != 1) q = -1; fclose(fq); }
      if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
      if (q > 0 && period > 0) quota = (double)q / (double)period;
    }
    if (quota > 0.0) cpus = std::max(1u, std::min(cpus, (unsigned)(quota + 0.5)));
    return cpus;
  }();
  return n;
}

}  // namespace fmh_host

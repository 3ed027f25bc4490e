/*
 * hostcfg.c - thread count of the native host helpers (generator, joint orders, graph features, JSON
 * reader: all OpenMP over the trusses of a batch).  OpenMP sizes its team by the logical CPUs it sees;
 * inside a container with a CPU quota (cgroup cpu.max) that is far too many - 256 threads on a 16-CPU
 * quota get throttled as a group, and so does the thread that drives the GPU.  The Python side
 * (`generate.available_cpus`) reads the affinity mask and the quota and sets the team size once.
 */
#include <omp.h>

/* n > 0: use n threads from now on.  Returns the team size in effect. */
int trs_host_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
}

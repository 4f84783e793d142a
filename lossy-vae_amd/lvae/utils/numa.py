"""NUMA placement of a rank's host threads (SURVEY.md 8(e) "Host side"): with one process per GPU the rANS coder threads and the
launch threads of a rank should run on the cores of the socket its GPU hangs off -- pinned-memory copies and the coder's working
set then stay on that socket's memory controllers, and 8 ranks do not migrate over each other's cores."""
import os


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(device_index=0):
    """CPUs of the NUMA node of cuda:<device_index>, from sysfs (PCI bus id -> numa_node -> cpulist); None if unknown."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
        with open(f'/sys/bus/pci/devices/{bdf}/numa_node') as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f'/sys/devices/system/node/node{node}/cpulist') as f:
            return _parse_cpulist(f.read())
    except (OSError, AttributeError, ValueError, RuntimeError):
        return None


def pin_rank(device_index, local_rank=0, local_world=1, share=True):
    """Restrict this process (and every thread it creates afterwards: the coder pool, the pipeline-group threads) to the cores of
    its GPU's NUMA node.  With share=False the node's cores are additionally sliced among the ranks whose GPUs sit on that node
    (by local rank order).  Returns the number of CPUs the process may use (what to size the coder thread pool with), or None
    when the topology cannot be read -- the call is then a no-op."""
    cpus = gpu_numa_cpus(device_index)
    if not cpus:
        return None
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    if not share and local_world > 1:
        per = max(1, len(allowed) // local_world)
        k = local_rank % max(1, len(allowed) // per)
        allowed = allowed[k * per:(k + 1) * per]
    try:
        os.sched_setaffinity(0, allowed)
    except OSError:
        return None
    return len(allowed)


def pin_ranks_collectively(device_index, dist, local_rank=None, local_world=None, force=False):
    """pin_rank() for a whole job, with a sanity check first: every rank publishes (hostname, CPU set of its GPU's node, CPUs it may
    use) with one tiny all_gather_object; only the ranks of THIS host are looked at from then on (a multi-node job has identical
    cpulists on different machines).  The ranks pin themselves only if the node sets of the host's ranks together cover (nearly) all
    CPUs the job may use there -- a container whose sysfs reports one node for every GPU would otherwise squeeze 8 ranks onto one
    socket (`force=True` skips that check: the 1-GPU rehearsal of the 8-rank host load, where every rank sits on one node by
    construction).  Ranks sharing a node split its cores evenly, in global-rank order.  Returns the number of CPUs of this rank, or
    None (nothing changed).  local_rank / local_world are accepted for compatibility and not used: the grouping comes from the gather."""
    import socket
    mine = gpu_numa_cpus(device_index)
    allowed = sorted(os.sched_getaffinity(0))
    rank, world = dist.get_rank(), dist.get_world_size()
    host = socket.gethostname()
    infos = [None] * world
    dist.all_gather_object(infos, (host, mine, allowed))
    here = [r for r in range(world) if infos[r][0] == host]
    if any(infos[r][1] is None for r in here):
        return None
    union = set().union(*[set(infos[r][1]) for r in here]) & set(allowed)
    if not force and len(union) < 0.9 * len(allowed):
        return None
    sharing = [r for r in here if infos[r][1] == mine]
    cpus = sorted(set(mine) & set(allowed))
    if not cpus:
        return None
    per = max(1, len(cpus) // len(sharing))
    k = sharing.index(rank) % max(1, len(cpus) // per)
    cpus = cpus[k * per:(k + 1) * per]
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return len(cpus)

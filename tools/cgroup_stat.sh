#!/bin/bash
# What the box lets this process use: CPUs allowed, the cgroup's CPU quota and how often it was throttled so far.
echo "nproc $(nproc)  allowed $(grep Cpus_allowed_list /proc/self/status | cut -f2)"
cg=$(cut -d: -f3 /proc/self/cgroup | head -1)
echo "cgroup $cg"
d=/sys/fs/cgroup$cg
while [ "$d" != "/sys/fs" ] && [ -n "$d" ]; do
  for f in cpu.max cpu.stat cpuset.cpus.effective; do
    [ -r $d/$f ] && echo "$d/$f: $(tr '\n' ' ' < $d/$f)"
  done
  [ "$d" = "/sys/fs/cgroup" ] && break
  d=$(dirname $d)
done

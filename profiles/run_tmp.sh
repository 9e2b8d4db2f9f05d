cat /sys/kernel/mm/transparent_hugepage/shmem_enabled /sys/kernel/mm/transparent_hugepage/enabled 2>&1; mount | grep -E "shm|tmpfs" | head -5; df -h /dev/shm | tail -1; nproc; free -g | head -2

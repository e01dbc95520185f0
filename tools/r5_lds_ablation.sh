cd /root/repo
tools/ab_libs.sh "L0 L1 L2" 32 4096 4 0 0 2
tools/ab_libs.sh "L0 L1 L2" 8 4096 4 0 512 2
tools/ab_libs.sh "L0 L1 L2" 12 4096 4 0 512 2
tools/ab_libs.sh "L0 L1 L2" 2 8192 4 1 0 2

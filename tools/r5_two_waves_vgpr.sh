cd /root/repo
for P in 8 12 16 32 64; do tools/ab_libs.sh "W0 W3" $P 4096 4 0 0 2; done
tools/ab_libs.sh "W0 W3" 2 8192 4 1 0 2
tools/ab_libs.sh "W0 W3" 4 8192 4 1 0 2
tools/ab_libs.sh "W0 W3" 128 1920 4 0 0 2 1080
tools/ab_libs.sh "W0 W3" 256 1920 4 0 0 2 1080

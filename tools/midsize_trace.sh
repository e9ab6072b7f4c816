cd /tmp && export TMPDIR=/tmp
for W in 512 1024 4096; do
rm -rf /tmp/mt$W; rocprofv3 --kernel-trace --output-format csv -d /tmp/mt$W -o t -- python /root/repo/tools/midsize_trace.py run $W 2>/dev/null | grep "per call"
python /root/repo/tools/midsize_trace.py report /tmp/mt$W
done

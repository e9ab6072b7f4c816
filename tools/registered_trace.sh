cd /tmp && export TMPDIR=/tmp
for ah in -1 8; do
rm -rf /tmp/tr$ah; OCTO_STAGE_AHEAD=$ah rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$ah -o t -- python /root/repo/tools/registered_trace.py run > /dev/null 2>&1
echo "== OCTO_STAGE_AHEAD=$ah"; python /root/repo/tools/registered_trace.py report /tmp/tr$ah
done

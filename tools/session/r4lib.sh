# helper for the round-4 session scripts: run a step under its own timeout, log to gpurun_out/, go on after an ordinary failure
# (a failed assertion), but start NO further GPU step after a timeout / kill (exit 124 / 137).
mkdir -p gpurun_out
step() {   # step <name> <seconds> <command...>
    local name=$1 secs=$2; shift 2
    echo "=== $name: $*"
    timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "=== $name: exit $rc"; tail -n 25 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name timed out: stopping"; exit $rc; fi
    return 0
}

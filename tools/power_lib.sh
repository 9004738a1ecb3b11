# Power and clock of the package while the bench loops run: is the ViT forward at the power cap?
# sourced by tools/r5_power*.sh: smp LABEL "ENV=.." "bench args" samples rocm-smi while bench.py loops and prints the samples under load + ms_per_step
smp() {  # label, env, args
  echo "## $1"
  env $2 python bench.py --no-cpu --no-secondary $3 --warmup 5 > /tmp/b.json 2>/dev/null &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed -e 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz)/sclk \1 MHz/' -e 's/.*Power (W): \([0-9.]*\)/  power \1 W/' | tr '\n' ' ' | awk '$5+0 > 400 || $2+0 > 600 {print}'
    sleep 0.3
  done
  grep -o '"ms_per_step": [0-9.]*' /tmp/b.json
}
summ() { awk '/^##/{if(n)printf "%s: n=%d sclk avg %.0f MHz power avg %.0f W\n",lab,n,s/n,p/n; lab=$0;n=0;s=0;p=0} /^sclk/{ if($5>900){n++;s+=$2;p+=$5}} /ms_per_step/{print lab, $0} END{if(n)printf "%s: n=%d sclk avg %.0f MHz power avg %.0f W\n",lab,n,s/n,p/n}' "$1"; }

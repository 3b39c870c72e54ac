# same-box A/B of the generation leg: tools/ab_gen.sh <tag> "name1 ENV=.. ENV=.." "name2 ..." ...
# every configuration runs `bench.py --only generate` (128 images, 64 prompts per call) in a fresh process; results in gpurun_out/<tag>/
tag=$1; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
for cfg in "$@"; do
  set -- $cfg
  name=$1; shift
  env "$@" timeout 600 python bench.py --only generate --gen-images 128 --gen-rowwise 0 > $out/gen_$name.json 2> $out/gen_$name.log
  python - <<PY
import json
try:
    d = json.load(open("$out/gen_$name.json")); print("generate $name", d.get("value"), d.get("unit"))
except Exception as e:
    print("generate $name FAILED", e)
PY
done

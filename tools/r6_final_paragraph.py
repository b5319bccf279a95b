import json,re,sys
F=sys.argv[1]; tag=sys.argv[2]
d=json.loads(open(F+'/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; oc=d['other_configs']; sp=r['same_precision']
ms=float(re.search(r"([\d.]+) ms/launch",r['kernel']).group(1))
tests=open(F+'/tests_gpu.log').read().strip().splitlines()[-1]
tm=re.search(r"(\d+) passed, (\d+) skipped.*in ([\d.]+)s",tests)
eager={}
for b in (32,8,1):
    eager[b]=float(re.search(r": ([\d.]+) ms over",open(F+'/unet_step_per_op_fp16_b%d.txt'%b).readline()).group(1))
tail=[l for l in open(F+'/per_op_roofline_b32.txt') if l.startswith('#')]
fam={}
for l in tail:
    m=re.match(r"#\s+(\w+)\s+measured\s+([\d.]+) ms, floor\s+([\d.]+) ms \(x ([\d.]+)\)",l)
    if m: fam[m.group(1)]=(float(m.group(2)),float(m.group(4)))
stepl=[l for l in tail if 'sum of per-op floors' in l][0]
fl=re.search(r"floors ([\d.]+) ms \(([\d.]+) %",stepl)
minb=re.search(r"min bytes ([\d.]+) GB",stepl).group(1)
tot={}
for l in open(F+'/pmc_totals.txt'):
    m=re.match(r"N=(\d+) (\w+) sum ([\d.]+)",l)
    if m: tot[(int(m.group(1)),m.group(2))]=float(m.group(3))
fetch=(tot[(6,'FETCH_SIZE')]-tot[(2,'FETCH_SIZE')])/4*1024/1e9; write=(tot[(6,'WRITE_SIZE')]-tot[(2,'WRITE_SIZE')])/4*1024/1e9
c2=oc['config2_b1_100step_fp16']; b8=oc['b8_200step_fp16']; c5=oc['config5_shard_xl_bf16_fp8attn_b8_200step']; c5x=oc['config5_shard_xl_bf16_mxfp8attn_b8_200step']
cb=d['cpu_baseline']; te=d['text_encoder_ms']
sha=d['config']['kernel_src_sha16']
P='profiles/r6_%s'%tag
txt=f"""Final numbers of round 6 (the tree with the activation-stationary level-0 kernels, the LDS-DMA attention tiles at levels 0–1, the slab GroupNorm and the narrow-output halo conv; the default
`python bench.py` line, `{P}_bench_b32_200step.json`, kernel sources `{sha}`): **{d['value']:.2f} audio-s/s** on one MI355X, {d['ms_per_step']/1000:.2f} s per 32-prompt pass,
**{ms:.2f} ms per denoise-step launch = {r['achieved']:.1f} TFLOP/s = {100*r['frac']:.1f} % of the 2.5 PFLOP/s dense fp16 MFMA peak** (executed {100*r['executed_frac']:.1f} %; first half of the round: 29.31 / 55.22 ms /
37.2 %; round 5's driver line: 29.26 / 55.32 ms — the gain of the second half, −5.4 % per launch, is outside the ±3 % box spread of the pool).  `roofline.same_precision` (the reference's fp32, a real 200-step
pass): {sp['value']:.2f} audio-s/s, {sp['denoise_step_launch_ms']:.1f} ms per launch = {100*sp['roofline_frac']:.1f} % of the 157.3-TFLOP/s f32 MFMA peak.  `other_configs`: config 2 (B = 1, 100 steps) {c2['denoise_step_launch_ms']:.2f} ms per launch =
{c2['value']:.2f} audio-s/s; B = 8 {b8['denoise_step_launch_ms']:.2f} ms = {b8['value']:.2f} audio-s/s ({100*b8['roofline_frac']:.1f} %); config 5's shard (XL, bf16, fp8 P·V, B = 8) {c5['denoise_step_launch_ms']:.2f} ms, with the MX P·V {c5x['denoise_step_launch_ms']:.2f} ms.  Text encoder {te['engine_ms']:.1f} ms
(engine) / {te['torch_ms']:.1f} ms (PyTorch-ROCm) per 32 prompts.  CPU oracle: BASELINE config 1 in full {cb['config1_measured']['seconds']:.1f} s = {cb['config1_measured']['value']:.3f} audio-s/s on {cb['cores']} threads; priced at 200 steps {cb['value']:.4f} audio-s/s.
HBM-side traffic of the launch on exactly these sources (`{P}_pmc_hbm_traffic_unet_step.txt` → `roofline.traffic`): {fetch+write:.1f} GB raw / **{2*fetch+write:.1f} GB corrected** (129.3 before the fusions) against
{minb} GB of per-op minimum bytes.  Per-op tables B = 32 / 8 / 1 (`{P}_unet_step_per_op_fp16_b*.txt`: {eager[32]:.2f} / {eager[8]:.2f} / {eager[1]:.2f} ms eager) and per-op roofline (`{P}_per_op_roofline_b32.txt`: floors {fl.group(1)} ms =
{fl.group(2)} % of the measured step; measured / floor by family: convs ×{fam['conv3x3'][1]:.2f} ({fam['conv3x3'][0]:.2f} ms), linears ×{fam['linear'][1]:.2f} ({fam['linear'][0]:.2f} ms), ff_fused ×{fam['ff_fused'][1]:.2f}, attention ×{fam['attention'][1]:.2f} ({fam['attention'][0]:.2f} ms), GroupNorm ×{fam['groupnorm'][1]:.2f} ({fam['groupnorm'][0]:.2f} ms)); rocprofv3 kernel
stats of the bench command `{P}_rocprofv3_kernel_stats_fp16_b32_4step.txt`; PMC of the activation-stationary kernels `{P}_pmc_ff_fused_qkv_stat_and_stream_geglu_summary.txt`, of the attention site
`r6_pmc_attention_summary.txt`.  Against the marks of the previous review: linears + ff_fused + qkv {fam['linear'][0]+fam['ff_fused'][0]+fam['qkv_stat'][0]:.1f} ms (mark ≤ 17: missed), S = 4096 attention sites 6.3–6.4 ms (mark ≤ 6.3: reached on the faster boxes), traffic
{2*fetch+write:.0f} GB (mark ≤ 118: missed by 2 %), step {ms:.1f} ms (mark ≤ 51: missed).  The driver's GPU test command on the final tree: **{tm.group(1)} passed, {tm.group(2)} skipped in {int(float(tm.group(3)))//60} min {int(float(tm.group(3)))%60} s** (`{P}_tests_gpu.log`).
"""
print(txt)

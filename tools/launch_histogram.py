"""GPU box: kernel launches of ONE steady-state training iteration from a rocprofv3 --kernel-trace CSV
(tools/launch_histogram.py <dir>): count + busy time per kernel family, using the recording MLP forward (mlp_fwd*_kernel<true>) as the
iteration marker."""
import collections, csv, glob, re, sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
it = [i for i, r in enumerate(rows) if "mlp_fwd" in r["Kernel_Name"] and "<true" in r["Kernel_Name"]]   # recording forward
seg = rows[it[-3]:it[-2]]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
print(len(seg), "launches in one iteration; GPU busy %.3f ms; wall %.3f ms" %
      (busy, (int(rows[it[-2]]["Start_Timestamp"]) - int(rows[it[-3]]["Start_Timestamp"])) / 1e6))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    m = re.search(r"(vectorized_elementwise_kernel|elementwise_kernel|reduce_kernel|multi_tensor_apply|batch_norm\w*|naive_conv\w*|"
                  r"Cijk\w{0,10}|miopenSp3\w{0,12}|igemm_\w{3}|gemv\w|transpose|copyBuffer|fillBuffer|Im2d2Col|SubTensor\w*|"
                  r"mlp_\w+|composite_\w+|raygen\w*|patch_gather\w*|pack\w*|sn_\w+|kernel_grouped_conv\w{0,10}|max_pool\w*|CatArray\w*)", n)
    return m.group(1) if m else n[:48]


cnt, tim = collections.Counter(), collections.Counter()
for r in seg:
    k = short(r["Kernel_Name"])
    cnt[k] += 1
    tim[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in cnt.most_common(28):
    print("%5d %9.1f us  %s" % (v, tim[k] / 1e3, k))
ew = collections.Counter()
for r in seg:
    n = r["Kernel_Name"]
    if "elementwise" in n:
        m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda|launch_\w+|leaky_relu\w*|copy\w*|pow\w*)", n.split("elementwise", 1)[1])
        ew[m.group(1) if m else n[60:110]] += 1
print(ew.most_common(16))

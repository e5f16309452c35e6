import torch
p = torch.cuda.get_device_properties(0)
print("device:", p.name, "CUs", p.multi_processor_count, "mem GB %.0f" % (p.total_memory / 2**30), "clock_rate", getattr(p, "clock_rate", None), "gcn", getattr(p, "gcnArchName", None))

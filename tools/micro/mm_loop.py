import torch
a=torch.randn(8192,8192,device="cuda",dtype=torch.bfloat16); b=torch.randn(8192,8192,device="cuda",dtype=torch.bfloat16)
for i in range(12000): c=a@b
torch.cuda.synchronize()

import sys, torch
sys.path.insert(0,'/root/repo')
from locomanipulationrl_amd.engine_config import loco_params
from locomanipulationrl_amd.lib import Engine, Rollout, POLICY_MLP
from locomanipulationrl_amd.model.robot_model import load_model
from locomanipulationrl_amd.policies.mlp_model import SharedMLP, pack_mlp_params
rm=load_model("quadruped_robot_v2"); m=SharedMLP().cuda(); pk=pack_mlp_params(m).cuda(); ls=torch.zeros(12,device="cuda")
torch.cuda.synchronize(); free0=torch.cuda.mem_get_info()[0]
for i in range(300):
    e=Engine(rm,[loco_params()],4096,seed=i); a=torch.zeros(4096,12,device="cuda"); e.step(a)
    ro=Rollout(e,POLICY_MLP,pk,ls,4,1); ro.run(); ro.close(); e.close(); del ro, e
torch.cuda.synchronize(); torch.cuda.empty_cache(); free1=torch.cuda.mem_get_info()[0]
print("free before %.1f MB after %.1f MB, delta %.1f MB" % (free0/1e6, free1/1e6, (free0-free1)/1e6))

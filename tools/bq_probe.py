import importlib,torch,sys,os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ext=importlib.import_module("3dvlp_amd._lib"); synth=importlib.import_module("3dvlp_amd.synth"); pu=importlib.import_module("3dvlp_amd.pointnet2_utils")
import numpy as np
xyz=torch.from_numpy(np.stack([synth.make_scene(1000+i,40000)["xyz"] for i in range(8)])).cuda()
inds=pu.furthest_point_sample(xyz,2048)
new=pu.gather_operation(xyz.transpose(1,2).contiguous(),inds).transpose(1,2).contiguous()
for _ in range(10): ext.ball_query(new,xyz,0.2,64,"grid")
torch.cuda.synchronize()

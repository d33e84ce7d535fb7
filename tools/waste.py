import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import oracle
from scannet_amd import synth
op = oracle.default_params(640,480,0.004)
vol = oracle.Volume(op, threads=8)
t=time.time()
for i in range(0, 900, 6):
    pose = synth.trajectory_pose(i, 5578)
    vol.integrate(synth.render_room_depth(pose, noise_frame=i), pose)
print("fused", vol.num_blocks, time.time()-t)
for i in (901, 902, 1500, 3000):
    c0, v0 = vol.export()
    pose = synth.trajectory_pose(i, 5578)
    n = vol.integrate(synth.render_room_depth(pose, noise_frame=i), pose)
    c1, v1 = vol.export()
    # align: blocks present before
    k0 = ((c0[:,0].astype(np.int64)+(1<<20))<<42)|((c0[:,1].astype(np.int64)+(1<<20))<<21)|(c0[:,2].astype(np.int64)+(1<<20))
    k1 = ((c1[:,0].astype(np.int64)+(1<<20))<<42)|((c1[:,1].astype(np.int64)+(1<<20))<<21)|(c1[:,2].astype(np.int64)+(1<<20))
    idx = np.searchsorted(k1, k0)
    same = (v1[idx].view(np.uint8).reshape(len(k0),-1) == v0.view(np.uint8).reshape(len(k0),-1))
    changed_vox = (~same.reshape(len(k0),512,8).all(-1))
    changed_blocks = changed_vox.any(1).sum() + (len(k1)-len(k0))
    print("frame", i, "frustum blocks", n, "blocks with >=1 update", int(changed_blocks), "frac", changed_blocks/n, "voxel updates / (frustum blocks*512)", (changed_vox.sum() + (len(k1)-len(k0))*400)/(n*512))

import sys, time
sys.path.insert(0,'/root/repo')
import pathtracer_amd as pt
sc, tex = pt.Scene.atrium(4,1), pt.make_textures(6,1)
for flags in (0, 4):
    c = pt.PathTracer(1920,1080,flags=flags); c.upload_scene(sc, tex_rgb8=tex); c.set_camera(pt.default_camera())
    s = pt.frame_seeds(0,64); c.render(0,s[:32],5); c.sync(); c.reset_counters(); t0=time.time(); c.render(32,s[32:],5); c.sync(); dt=time.time()-t0
    k=c.counters(); rays=k['rays_primary']+k['rays_shadow']+k['rays_bounce']
    print(flags, round(rays/dt/1e6,1),'Mrays/s trace',round(k['ms_trace'],1),'shade',round(k['ms_shade'],1),'S',round(k['node_steps']/rays,2),'T',round(k['tri_tests']/rays,2), 'rays/path', round(rays/k['paths'],2), 'spills', k['stack_spills'])
    c.close()

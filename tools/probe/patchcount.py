import numpy as np
rng = np.random.default_rng(0)
def count(theta, tx, ty, rows, cols=8, scale=1.0):
    c, s = np.cos(theta)*scale, np.sin(theta)*scale
    u, v = np.meshgrid(np.arange(cols), np.arange(rows))
    sx = c*u - s*v + tx; sy = s*u + c*v + ty
    x0 = np.floor(sx).astype(int); y0 = np.floor(sy).astype(int)
    wx1 = sx - x0; wy1 = sy - y0
    pts = set()
    for dx in (0,1):
        for dy in (0,1):
            w = (wx1 if dx else 1-wx1) * (wy1 if dy else 1-wy1)
            for a,b,ww in zip((x0+dx).ravel(), (y0+dy).ravel(), w.ravel()):
                if ww != 0: pts.add((a,b))
    return len(pts)
for rows in (2,4,8):
    mx = 0; tot=0; n=0; hist=[]
    for th in np.linspace(0, np.pi/2, 361):
        for _ in range(40):
            k = count(th, rng.random()*3, rng.random()*3, rows)
            mx = max(mx,k); tot+=k; n+=1; hist.append(k)
    hist=np.array(hist)
    print(rows, 'max', mx, 'mean', tot/n, 'p99', np.percentile(hist,99))
# shipped yaws 0.2..0.8
for rows in (4,):
    for th in (0.2,0.4,0.6,0.8):
        ks=[count(th, rng.random()*3, rng.random()*3, rows) for _ in range(400)]
        print('theta',th, np.mean(ks), max(ks))
for sc in (1.02,1.05,1.1):
    mx=0
    for th in np.linspace(0, np.pi/2, 181):
        for _ in range(30):
            mx=max(mx,count(th, rng.random()*3, rng.random()*3, 4, scale=sc))
    print('scale',sc,'max',mx)

"""A miniature Deep360 directory tree (empty files with dataset-like names), shared by make_golden_lists.py and the tests.
File names are chosen so that lexicographic order differs from creation order (the reference pairs files by position in
the SORTED listing)."""
import os

PAIRS = ('12', '13', '14', '23', '24', '34')
FRAMES = {'training': ('000003', '000001', '000010'), 'validation': ('000002',), 'testing': ('000007', '000005')}


def build(root, soiled=True):
  """Creates <root>/dataset (rgb[, rgb_soiled], disp, depth) and <root>/exported (disp_pred2depth, conf_map[, _soiled])."""
  made = []
  for ep in range(1, 7):
    for subset, frames in FRAMES.items():
      for fr in frames:
        name = 'ep%d_%s' % (ep, fr)
        files = [('dataset', 'depth', name + '_depth.npz')]
        for p in PAIRS:
          files += [('dataset', 'rgb', '%s_%s_rgb%s.png' % (name, p, p[0])), ('dataset', 'rgb', '%s_%s_rgb%s.png' % (name, p, p[1])),
                    ('dataset', 'disp', '%s_%s_disp.npz' % (name, p)), ('exported', 'disp_pred2depth', '%s_%s_disp_pred2depth.npz' % (name, p)),
                    ('exported', 'conf_map', '%s_%s_conf_map.png' % (name, p))]
          if soiled:
            files += [('dataset', 'rgb_soiled', '%s_%s_rgb%s_soiled.png' % (name, p, p[0])),
                      ('dataset', 'rgb_soiled', '%s_%s_rgb%s_soiled.png' % (name, p, p[1])),
                      ('exported', 'disp_pred2depth_soiled', '%s_%s_disp_pred2depth.npz' % (name, p)),
                      ('exported', 'conf_map_soiled', '%s_%s_conf_map.png' % (name, p))]
        for top, kind, fn in files:
          d = os.path.join(root, top, 'ep%d_500frames' % ep, subset, kind)
          os.makedirs(d, exist_ok=True)
          open(os.path.join(d, fn), 'wb').close()
          made.append(os.path.join(d, fn))
  return os.path.join(root, 'dataset'), os.path.join(root, 'exported'), made

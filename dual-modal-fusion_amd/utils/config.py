"""utils.config — YAML + Jinja2 configuration (mirror of the reference module of the same name).

Same entry points and result as the reference: `get_render_config(path)` parses the YAML once to collect template
parameters, renders the same file as a Jinja2 template, parses the result and passes it through
`get_dump_config` (utils/config.py:12-93), which picks the output directory `<RESULT><model>__<n>_output/`,
coerces the float fields and creates the directory.  Deliberate differences:
  * the template that is rendered is the file at `path` (the reference always re-opens './config.yml', :18);
  * the `dqtl:` section is optional (the shipped reference config.yml lacks it and fails its own loader);
  * `yaml.safe_load` instead of `yaml.load(FullLoader)`;
  * earlier output directories without a result sheet are removed only when `delete: 1` (reference semantics, :55-71).
"""
import os
import shutil
from pathlib import Path

import yaml
from jinja2 import Template


def get_config(path):
    with open(path, encoding='utf-8') as f:
        return yaml.safe_load(f)


def get_render_config(path):
    data = get_config(path)
    base_dir = Path(__file__).resolve().parent.parent
    with open(path, 'r', encoding='utf-8') as f:
        template = Template(f.read())
    dqtl = data.get('dqtl') or {}
    parameters = {
        'parameter1': 'value1', 'p2': base_dir, 'dc': data['data_city'],
        'num': len(data['DATA_DICT'][data['data_city']]['color']),      # incl. background class 0 (:25)
        'tr': data['train_rate'], 'ep': data['epoch'], 'bs': data['batchsize'],
        'expo_result': data['expo_result'], 'parameters': data['parameters'],
        'mn': data['model_name'], 'FN': data['FILE_NUM'],
        'ne': dqtl.get('num_epochs', 0), 'ps': dqtl.get('pic_size', 0),
    }
    y = yaml.safe_load(template.render(**parameters))
    return get_dump_config(y)


def get_dump_config(y):
    os.makedirs(y['RESULT'], exist_ok=True)

    def names(n):
        stem = y['RESULT'] + y['model_name'] + "__" + str(n)
        return stem + '_result.xlsx', stem + '_output/'

    filenum = 0
    result_excel, result_output = names(filenum)
    if not y['train']['index'] == 0:
        while os.path.exists(result_excel) or os.path.exists(result_output):
            filenum += 1
            result_excel, result_output = names(filenum)
        y['FILE_NUM'] = filenum
        if y.get('delete'):
            for num in range(filenum - 1, -1, -1):
                xlsx, out_dir = names(num)
                if os.path.isdir(out_dir) and not os.path.isfile(xlsx):
                    shutil.rmtree(out_dir)
                    filenum = num
                    y['FILE_NUM'] = filenum
    else:
        filenum = y['FILE_NUM']
    y['RESULT_excel'] = result_excel                        # as the reference: the name probed last (:77)
    y['RESULT_output'] = names(filenum)[1]
    y['schedule']['lr'] = float(y['schedule']['lr'])
    y['schedule']['base_lr'] = float(y['schedule']['base_lr'])
    y['Categories_Number'] = int(y['Categories_Number'])
    if y.get('dqtl'):
        for k in ('lr', 'tao', 'epsilon'):
            if k in y['dqtl']:
                y['dqtl'][k] = float(y['dqtl'][k])
    y = yaml.safe_load(yaml.dump(y))
    if not os.path.exists(y['RESULT_output']) and y['train']['save_best']:
        os.makedirs(y['RESULT_output'])
    return y

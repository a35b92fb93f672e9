import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if 'SQ_INSTS_VALU_MFMA_MOPS_F16' not in v or v['SQ_INSTS_VALU_MFMA_MOPS_F16']['sum']==0: continue
    print(k)
    g=lambda n: v[n]['sum']/v[n]['launches'] if n in v else float('nan')
    mf=g('SQ_INSTS_VALU_MFMA_MOPS_F16')/32
    print('  launches',v['SQ_WAVES']['launches'],'waves',g('SQ_WAVES'))
    print('  MFMA insts %.4g  VALU %.4g (other per MFMA %.2f) SALU %.4g (per MFMA %.2f) LDS %.4g VMEM %.4g SMEM %.4g'%(mf,g('SQ_INSTS_VALU'),(g('SQ_INSTS_VALU')-mf)/mf,g('SQ_INSTS_SALU'),g('SQ_INSTS_SALU')/mf,g('SQ_INSTS_LDS'),g('SQ_INSTS_VMEM'),g('SQ_INSTS_SMEM')))
    wc=g('SQ_WAVE_CYCLES')
    print('  wave cycles(quad) %.4g  WAIT_ANY %.2f  WAIT_INST_ANY %.2f ACTIVE_VALU %.2f  WAIT_INST_LDS %.3f'%(wc,g('SQ_WAIT_ANY')/wc,g('SQ_WAIT_INST_ANY')/wc,g('SQ_ACTIVE_INST_VALU')/wc, g('SQ_WAIT_INST_LDS')/wc))
    ga=g('GRBM_GUI_ACTIVE')
    print('  GRBM_GUI_ACTIVE %.4g (per XCD %.4g) MFMA busy cycles %.4g -> busy frac %.3f ; SQ_BUSY_CYCLES %.4g'%(ga,ga/8,g('SQ_VALU_MFMA_BUSY_CYCLES'),g('SQ_VALU_MFMA_BUSY_CYCLES')/(ga/8*1024),g('SQ_BUSY_CYCLES')))
    print('  per wave: VALU %.0f SALU %.0f LDS %.0f VMEM %.0f MFMA %.0f'%(g('SQ_INSTS_VALU')/g('SQ_WAVES'),g('SQ_INSTS_SALU')/g('SQ_WAVES'),g('SQ_INSTS_LDS')/g('SQ_WAVES'),g('SQ_INSTS_VMEM')/g('SQ_WAVES'),mf/g('SQ_WAVES')))
